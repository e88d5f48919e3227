"""Choice <-> token encoding and the mutation operator of the evolutionary search (reference: nasrec/searcher/tokenizer.py).

Host-side integer code; what matters for equivalence is (a) the token layout — it is the identity of an architecture in the
search history — and (b) the ORDER of the global `np.random` draws in `mutate_spec` / `generate_random_choice`, which decides
which children a seeded search visits.  Both are pinned to traces recorded from the reference's Tokenizer
(tests/golden/tokenizer.json).  The reference's `tokenize` ends in `np.asarray(..., dtype=np.int)` (tokenizer.py:182), an alias
numpy removed in 1.24; the plain `int` it stood for is used here."""
from copy import deepcopy
from typing import Any, Dict, List

import numpy as np


class Tokenizer(object):
    def __init__(self, num_blocks: int, ops_config: Any):
        self._num_blocks = num_blocks
        self._ops_config = ops_config
        per_block = ops_config if isinstance(ops_config, list) else [ops_config] * num_blocks
        self._per_block = per_block
        self._dense_code = [{d: i for i, d in enumerate(c["dense_node_dims"])} for c in per_block]
        self._sparse_code = [{d: i for i, d in enumerate(c["sparse_node_dims"])} for c in per_block]

    def _bits(self, members, n) -> List[int]:
        members = [int(m) for m in np.asarray(members).reshape(-1).tolist()]
        return [1 if i in members else 0 for i in range(n)]

    def tokenize(self, choice: Dict[Any, Any]) -> np.ndarray:
        """tokenizer.py:153-182: per block 4 x [num_blocks] connection bits; then per block [num_nodes] activity bits, the index
        of the dense width, the index of the sparse width, and two one-hot pairs (dense_sparse_interact, deep_fm)"""
        enc: List[int] = []
        for mac in choice["macro"]:
            for key in ("dense_idx", "sparse_idx", "dense_left_idx", "dense_right_idx"):
                enc += self._bits(mac[key], self._num_blocks)
        for i, mic in enumerate(choice["micro"]):
            cfg = self._per_block[i]
            enc += self._bits(mic["active_nodes"], cfg["num_nodes"])
            enc.append(self._dense_code[i][int(mic["dense_in_dims"])])
            enc.append(self._sparse_code[i][int(mic["sparse_in_dims"])])
            enc += [1, 0] if int(mic["dense_sparse_interact"]) == 0 else [0, 1]
            enc += [1, 0] if int(mic["deep_fm"]) == 0 else [0, 1]
        return np.asarray(enc, dtype=int)

    def hash_token(self, token) -> str:
        return "".join(str(x) for x in token)

    # ---- random draws (global np.random, reference order) --------------------------------------------------------------------
    def _draw_macro(self, block_idx: int) -> Dict[str, list]:
        """tokenizer.py:196-227 / 280-308: at most 4 inputs per connection type, one (left, right) pair"""
        n_dense = 1 + np.random.choice(min(4, block_idx + 1))
        n_sparse = 1 + np.random.choice(min(4, block_idx + 1))
        pair = np.random.choice(block_idx + 1, 2)
        return {"dense_idx": np.random.choice(block_idx + 1, n_dense, replace=False).reshape(-1).tolist(),
                "sparse_idx": np.random.choice(block_idx + 1, n_sparse, replace=False).reshape(-1).tolist(),
                "dense_left_idx": pair[:1].reshape(-1).tolist(), "dense_right_idx": pair[1:].reshape(-1).tolist()}

    def _draw_micro(self, block_idx: int) -> Dict[str, Any]:
        """tokenizer.py:247-260 / 326-339: one dense node + one sparse node, redrawn while it is the all-zero pair"""
        cfg = self._per_block[block_idx]
        while True:
            mic = {"active_nodes": sorted([np.random.choice(cfg["dense_nodes"])] + [np.random.choice(cfg["sparse_nodes"])]),
                   "dense_in_dims": np.random.choice(cfg["dense_node_dims"]), "sparse_in_dims": np.random.choice(cfg["sparse_node_dims"]),
                   "dense_sparse_interact": np.random.choice([0, 1]), "deep_fm": np.random.choice([0, 1])}
            if mic["active_nodes"] != cfg["zero_nodes"]:
                return mic

    def mutate_spec(self, choice: Dict[Any, Any]) -> Dict[Any, Any]:
        """tokenizer.py:188-263: resample ONE field of ONE block (a whole fresh macro / micro choice is drawn, one key of it kept)"""
        block_idx = np.random.choice(self._num_blocks)
        level = "macro" if np.random.random() > 0.5 else "micro"
        out = deepcopy(choice)
        if level == "macro":
            fresh = self._draw_macro(block_idx)
            key = np.random.choice(["dense_idx", "sparse_idx", "dense_left_idx", "dense_right_idx"])
            out["macro"][block_idx][key] = deepcopy(fresh[key])
        else:
            fresh = self._draw_micro(block_idx)
            key = np.random.choice(["active_nodes", "dense_in_dims", "sparse_in_dims", "dense_sparse_interact", "deep_fm"])
            out["micro"][block_idx][key] = fresh[key]
        return out

    def generate_random_choice(self) -> Dict[str, list]:
        """tokenizer.py:265-342 (the reference draws one unused block index first)"""
        np.random.choice(self._num_blocks)
        choice = {"macro": [], "micro": []}
        for b in range(self._num_blocks):
            mac = self._draw_macro(b)
            mic = self._draw_micro(b)
            choice["macro"].append(mac)
            choice["micro"].append(mic)
        return choice
