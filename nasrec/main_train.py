"""`python -u nasrec/main_train.py ...` (the command line of scripts/eval_best_model/*.sh) -> the engine-side harness"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nasrec_amd.main_train import build_parser, get_model, main, train_and_eval_one_model  # noqa: E402,F401

if __name__ == "__main__":
    main(build_parser().parse_args())
