#!/usr/bin/env python3
"""Golden traces of the search control plane's only arithmetic — the Tokenizer (nasrec/searcher/tokenizer.py:30-342) — recorded
by IMPORTING the real reference:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_search.py

For several (search space, num_blocks, seed): `generate_random_choice()` draws, chains of `mutate_spec()`, and the token /
hash of every choice.  `Tokenizer.tokenize` ends in `np.asarray(..., dtype=np.int)` (tokenizer.py:182), an alias numpy removed in
1.24; the generator restores the alias (`np.int = int`, what it always meant) so that the reference's own code produces the
tokens."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.setup()

import numpy as np  # noqa: E402

if not hasattr(np, "int"):
    np.int = int  # the removed alias the reference still spells out

from nasrec.searcher.tokenizer import Tokenizer  # noqa: E402  the REAL reference
from nasrec.supernet.supernet import ops_config_lib  # noqa: E402

_refimport.assert_reference_modules()
OUT = os.environ.get("GOLDEN_OUT", HERE)


def clean(v):
    if isinstance(v, dict):
        return {k: clean(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [clean(x) for x in v]
    if isinstance(v, np.ndarray):
        return clean(v.tolist())
    if isinstance(v, np.generic):
        return v.item()
    return v


def main():
    traces = []
    for space, nb in (("xlarge", 7), ("autoctr", 7), ("xlarge-zeros", 3)):
        for seed in (3, 17):
            tok = Tokenizer(num_blocks=nb, ops_config=ops_config_lib[space])
            np.random.seed(seed)
            seq = []
            choice = tok.generate_random_choice()
            seq.append(dict(op="generate", choice=clean(choice), token=clean(tok.tokenize(choice)), hash=tok.hash_token(tok.tokenize(choice))))
            for _ in range(12):
                choice = tok.mutate_spec(choice)
                t = tok.tokenize(choice)
                seq.append(dict(op="mutate", choice=clean(choice), token=clean(t), hash=tok.hash_token(t)))
            choice = tok.generate_random_choice()
            seq.append(dict(op="generate", choice=clean(choice), token=clean(tok.tokenize(choice)), hash=tok.hash_token(tok.tokenize(choice))))
            traces.append(dict(space=space, num_blocks=nb, seed=seed, seq=seq))
    json.dump(traces, open(os.path.join(OUT, "tokenizer.json"), "w"))
    print("wrote tokenizer.json", len(traces), "traces,", sum(len(t["seq"]) for t in traces), "choices")


if __name__ == "__main__":
    main()
