"""The search control plane on CPU: the Tokenizer against traces recorded from the reference's own Tokenizer
(tests/golden/tokenizer.json: token layout, hash, and the global-np.random draw order of generate_random_choice / mutate_spec),
and the Searcher's population bookkeeping (random search, regularized evolution) with a deterministic stand-in scorer."""
import argparse
import json
import os

import numpy as np

from helpers import GOLDEN
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.searcher.searcher import Searcher
from nasrec_amd.searcher.tokenizer import Tokenizer


def _clean(v):
    return json.loads(json.dumps(v, default=lambda o: o.tolist() if hasattr(o, "tolist") else o.item()))


def test_tokenizer_reproduces_the_reference_traces():
    traces = json.load(open(os.path.join(GOLDEN, "tokenizer.json")))
    assert len(traces) >= 6
    for tr in traces:
        tok = Tokenizer(num_blocks=tr["num_blocks"], ops_config=ops_config_lib[tr["space"]])
        np.random.seed(tr["seed"])
        choice = None
        for rec in tr["seq"]:
            choice = tok.generate_random_choice() if rec["op"] == "generate" else tok.mutate_spec(choice)
            assert _clean(choice) == rec["choice"], (tr["space"], tr["seed"], rec["op"])
            t = tok.tokenize(choice)
            assert t.tolist() == rec["token"]
            assert tok.hash_token(t) == rec["hash"]
            # a recorded choice re-encodes to the same token (JSON round trip: plain ints instead of numpy scalars)
            assert tok.tokenize(rec["choice"]).tolist() == rec["token"]


def _fake_eval(model, args, checkpoint):
    """deterministic 'quality' of a choice: no engine, no data — exercises the search bookkeeping only"""
    ch = _clean(model.choice)
    s = sum(sum(m["dense_idx"]) + 3 * sum(m["sparse_idx"]) for m in ch["macro"]) + sum(mi["dense_in_dims"] % 7 + mi["sparse_in_dims"] % 5 for mi in ch["micro"])
    return {"choice": ch, "test_acc": 0.5 + (s % 13) / 100.0, "test_auroc": 0.6 + (s % 11) / 100.0, "test_loss": 0.4 + (s % 17) / 100.0}


class _FakeModel:
    """what _create_model_train_and_get_results needs from a SuperNet when the scorer never runs it"""

    def __init__(self, tok):
        self.choice = tok.generate_random_choice()

    def configure_choice(self, c):
        self.choice = c


def test_searcher_population_bookkeeping(monkeypatch):
    from nasrec_amd.searcher import searcher as S
    from nasrec_amd.searcher import searcher_utils as SU
    args = argparse.Namespace(num_blocks=3, config="autoctr", dataset="kdd", use_layernorm=1, ckpt_path=None, gpu=None)
    tok = Tokenizer(3, ops_config_lib["autoctr"])
    monkeypatch.setattr(SU, "build_supernet", lambda a, t=None: _FakeModel(tok))
    # run the workers inline (no processes): the helper is the unit under test, process spawning is covered on the GPU
    def inline(self, choices, on_cpu, ckpt_holder, kwargs):
        out = []
        for job_id, ch in enumerate(choices):
            rd = {}
            a = argparse.Namespace(**vars(self._args), deterministic_workers=True)
            SU.create_model_train_and_get_results_helper(a, SU.get_device_id(job_id, on_cpu), self._eval_fn, self._tokenizer, ch, rd, ckpt_holder, kwargs)
            out += [rd[k] for k in sorted(rd)]
        return out
    monkeypatch.setattr(S.Searcher, "_run_jobs", inline)
    np.random.seed(5)
    s = Searcher(_fake_eval, args)
    top = s.random_search_from_supernet(budget=9, top_k=4, num_parallel_workers=1, on_cpu=True, sorted=True)
    assert len(s.all_results) == 9 and len(top) == 4
    losses = [r["test_loss"] for r in top]
    assert losses == sorted(losses) and losses[0] == min(r["test_loss"] for r in s.all_results)
    assert all(set(r) == {"choice", "test_acc", "test_auroc", "test_loss", "hash_token"} for r in s.all_results)
    best_auc = s._sort_results_with_criterion(s.all_results, "test_auroc")
    assert best_auc[0]["test_auroc"] == max(r["test_auroc"] for r in s.all_results)
    np.random.seed(6)
    hist = s.regularized_evolution_from_supernet(n_generations=3, n_childs=4, init_population=6, sample_size=3, num_parallel_workers=1,
                                                 on_cpu=True, top_k=2)
    assert len(hist) == 3 * 2  # top_k per generation
    hashes = [h["hash_token"] for h in hist]
    assert all(isinstance(h, str) and set(h) <= set("0123456789") for h in hashes)
    for h in hist:  # children are valid choices of the space: they re-tokenize to their own hash
        assert tok.hash_token(tok.tokenize(h["choice"])) == h["hash_token"]
