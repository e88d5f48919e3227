"""`python -u nasrec/train_supernet.py ...` (scripts/train_supernet/*.sh) -> the engine-side harness"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nasrec_amd.train_supernet import build_parser, main, train_and_eval_one_model  # noqa: E402,F401

if __name__ == "__main__":
    main(build_parser().parse_args())
