"""The training harness (nasrec_amd/main_train.py, train_supernet.py, utils/train_utils.py — SURVEY §8f-1) end to end on the
GPU: the command line of the published recipe on TSV shards, the reference's log / checkpoint artefacts, and the fused
engine step against the operator-by-operator torch route through the same loop."""
import copy
import json
import os
import pickle

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from nasrec_amd import main_train as MT
from nasrec_amd import train_supernet as TS
from nasrec_amd.utils import train_utils as TU

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_autoctr_best_1shot.json")


def _shards(tmp_path, ds="criteo-kaggle", repeat=3):
    z = np.load(os.path.join(GOLDEN, "datapipes.npz"), allow_pickle=False)
    root = tmp_path / "data"
    for s in range(2):
        d = root / ("shard-%d" % s)
        d.mkdir(parents=True)
        for name in ("trainval.txt", "train.txt", "test.txt"):
            src = "trainval.txt" if name == "train.txt" else name
            (d / name).write_text("\n".join([str(z["%s/shard-%d/%s" % (ds, s, src)])] * repeat) + "\n")
    return str(root)


def test_main_train_recipe_runs_and_writes_the_reference_artefacts(tmp_path, capsys):
    root, logdir = _shards(tmp_path), str(tmp_path / "logs")
    args = MT.build_parser().parse_args([
        "--root_dir", root, "--net", "supernet-config", "--supernet_config", CFG, "--num_epochs", "1", "--learning_rate", "0.1",
        "--train_batch_size", "8", "--test_batch_size", "16", "--wd", "0", "--logging_dir", logdir, "--gpu", "0", "--test_interval", "4",
        "--display_interval", "2", "--train_limit", "96"])
    torch.manual_seed(0)
    logs = MT.main(args)
    out = capsys.readouterr().out
    assert "FLOPS:" in out and "Epoch: 0 L2: 0.000000 loss:" in out and "Learning rate: 1e-08" in out and "Test Acc:" in out
    assert len(logs) == 1
    log = logs[0]
    assert set(log) == {"train_loss", "train_AUROC", "train_Accuracy", "test_loss", "test_AUROC", "test_Accuracy", "epoch", "iters"}
    assert log["iters"] == [0, 2, 4, 6, 8, 10, 11] and len(log["test_loss"]) == 4  # steps 0, 4, 8 and the last one (11)
    assert all(np.isfinite(v) for v in log["train_loss"] + log["test_loss"])
    assert pickle.load(open(os.path.join(logdir, "train_test_logs.pickle"), "rb")) == logs
    ck = torch.load(os.path.join(logdir, "supernet-config_checkpoint.pt"), map_location="cpu")
    assert set(ck) == {"model_state_dict", "optimizer_state_dict"}
    model = MT.get_model(args)
    assert list(ck["model_state_dict"])[:26] == ["_embedding.%d.weight" % f for f in range(26)]
    osd = ck["optimizer_state_dict"]
    assert len(osd["state"]) == len(ck["model_state_dict"])  # every parameter carries its Adagrad sum
    assert all(float(s["step"]) == 12.0 for s in osd["state"].values())
    assert any(float(s["sum"].abs().sum()) > 0 for s in osd["state"].values())
    del model


def test_fused_engine_step_equals_the_torch_route_through_the_same_loop(tmp_path):
    root = _shards(tmp_path)
    args = MT.build_parser().parse_args([
        "--root_dir", root, "--net", "supernet-config", "--supernet_config", CFG, "--learning_rate", "0.05", "--train_batch_size", "8",
        "--test_batch_size", "16", "--wd", "0", "--logging_dir", str(tmp_path / "l"), "--gpu", "0", "--train_limit", "48"])
    from nasrec_amd.utils.data_pipes import make_loaders
    train_loader, test_loader = make_loaders(args)
    torch.manual_seed(1)
    base = MT.get_model(args).to(0)
    with torch.no_grad():
        TU.warmup_model(base, train_loader, 0)
    base.apply(TU.init_weights)
    results = []
    for use_engine in (None, False):
        model = copy.deepcopy(base)
        opt = MT.build_optimizer("adagrad", model, args.learning_rate)
        sched = MT.build_lr_scheduler("constant", opt, 6, 2, args.learning_rate)
        logs = TU.train_and_test_one_epoch(model, 0, opt, sched, train_loader, test_loader, torch.nn.BCEWithLogitsLoss(),
                                           lambda m: TU.get_l2_loss(m, 0.0, None, gpu=0), 8, 0, display_interval=1, test_interval=100,
                                           max_train_steps=6, grad_clip_value=5.0, use_engine_step=use_engine)
        torch.cuda.synchronize()
        results.append((logs, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                        {i: s["sum"].detach().cpu().clone() for i, s in enumerate(opt.state_dict()["state"].values())}))
    (la, pa, sa), (lb, pb, sb) = results
    assert np.allclose(la["train_loss"], lb["train_loss"], rtol=1e-5, atol=1e-6)
    assert la["iters"] == lb["iters"] == [0, 1, 2, 3, 4, 5]
    for k in pa:
        assert torch.allclose(pa[k], pb[k], rtol=0, atol=2e-5), k
    for i in sa:
        assert torch.allclose(sa[i], sb[i], rtol=1e-4, atol=1e-7), i


def test_train_supernet_cli_on_synthetic_batches(tmp_path, capsys):
    args = TS.build_parser().parse_args([
        "--dataset", "kdd", "--root_dir", "synthetic:steps=6,test_steps=1,seed=3", "--logging_dir", str(tmp_path / "sn"), "--config", "autoctr",
        "--num_blocks", "3", "--use_layernorm", "1", "--strategy", "default", "--anypath_choice", "binomial-0.5", "--supernet_training_steps", "2",
        "--train_batch_size", "16", "--test_batch_size", "16", "--train_limit", "64", "--learning_rate", "0.05", "--display_interval", "2",
        "--gpu", "0"])
    np.random.seed(0)
    torch.manual_seed(0)
    logs = TS.main(args)
    out = capsys.readouterr().out
    assert "Logging in directory:" in out
    d = os.path.join(str(tmp_path / "sn"), "supernet_3blocks_layernorm1_default-binomial-0.5_lr0.05_supernetwarmup_2")
    assert os.path.exists(os.path.join(d, "supernet_checkpoint.pt")) and os.path.exists(os.path.join(d, "train_test_logs.pickle"))
    assert len(logs[0]["test_loss"]) == 1 and np.isfinite(logs[0]["test_loss"][0])  # test only at the last step


def test_finetune_last_layer_only_through_the_harness_leaves_everything_else_untouched(tmp_path):
    """eval_subnet_from_supernet.py:101-198: set_mode_to_finelune_last_only() + Adagrad(model.parameters(), eps=1e-2) + clip 5 through
    train_and_test_one_epoch.  The fused engine step would train the whole arena, so the loop must take the torch route on its
    own (use_engine_step=None) and only `_final` may move — every other parameter stays bit-identical."""
    root = _shards(tmp_path)
    args = MT.build_parser().parse_args([
        "--root_dir", root, "--net", "supernet-config", "--supernet_config", CFG, "--learning_rate", "0.05", "--train_batch_size", "8",
        "--test_batch_size", "16", "--wd", "0", "--logging_dir", str(tmp_path / "l"), "--gpu", "0", "--train_limit", "48"])
    from nasrec_amd.utils.data_pipes import make_loaders
    train_loader, test_loader = make_loaders(args)
    torch.manual_seed(2)
    model = MT.get_model(args).to(0)
    with torch.no_grad():
        TU.warmup_model(model, train_loader, 0)
    model.apply(TU.init_weights)
    model.set_mode_to_finelune_last_only()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt = torch.optim.Adagrad(model.parameters(), lr=0.05, eps=1e-2)
    assert TU._fused_step_applies(model, opt, lambda m: TU.get_l2_loss(m, 0.0, None, gpu=0), False) is False
    sched = MT.build_lr_scheduler("constant", opt, 6, 2, 0.05)
    sched.step(epoch=-1)  # eval_subnet_from_supernet.py:179
    TU.train_and_test_one_epoch(model, 0, opt, sched, train_loader, test_loader, torch.nn.BCEWithLogitsLoss(),
                                lambda m: TU.get_l2_loss(m, 0.0, None, gpu=0), 8, 0, display_interval=2, test_interval=100,
                                max_train_steps=4, grad_clip_value=5.0)
    torch.cuda.synchronize()
    after = model.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    assert sorted(moved) == ["_final.bias", "_final.weight"], moved
    # an optimizer over a parameter subset must not take the fused step either
    model.set_mode_to_normal_mode()
    sub = torch.optim.Adagrad(model.get_dense_parameters(), lr=0.05, eps=1e-2)
    assert TU._fused_step_applies(model, sub, lambda m: TU.get_l2_loss(m, 0.0, None, gpu=0), False) is False
    full = torch.optim.Adagrad(model.parameters(), lr=0.05, eps=1e-2)
    assert TU._fused_step_applies(model, full, lambda m: TU.get_l2_loss(m, 0.0, None, gpu=0), False) is True


def test_search_candidates_are_scored_by_last_layer_finetuning(monkeypatch):
    """eval_subnet_from_supernet.py's loop on the engine: random candidates of a (capped-table) KDD autoctr supernet, each scored by
    `finetune_and_eval_one_model` from a shared checkpoint — full-path warm-up, pinned path, last layer only.  Workers run inline
    here (a process that holds the GPU must not be duplicated; the CLI's parent process never touches the GPU and spawns one
    worker per device).  Every candidate returns finite scores, a distinct architecture hash, and leaves everything but `_final`
    bit-identical to the checkpoint."""
    import argparse
    from nasrec_amd import eval_subnet_from_supernet as E
    from nasrec_amd.searcher import searcher as S
    from nasrec_amd.searcher import searcher_utils as SU
    from nasrec_amd.utils.config import DATASETS
    tables = [min(n, 1000) for n in DATASETS["kdd"]["tables"]]
    args = E.build_parser().parse_args([
        "--dataset", "kdd", "--root_dir", "synthetic:steps=8,test_steps=2,seed=5,cap=1000", "--logging_dir", "/tmp/nasrec_search_test",
        "--config", "autoctr", "--num_blocks", "3", "--use_layernorm", "1", "--max_train_steps", "4", "--max_eval_steps", "2",
        "--train_batch_size", "64", "--test_batch_size", "64", "--method", "random", "--random_budget", "3", "--learning_rate", "0.05",
        "--display_interval", "2", "--test_only_at_last_step", "1"])  # as scripts/run_ea/*.sh: one test per candidate
    args.num_embeddings = tables
    # the supernet checkpoint every candidate starts from
    torch.manual_seed(3)
    base = SU.build_supernet(args, tables).to(0)
    int_x = torch.zeros(4, 3, device="cuda")
    cat_x = torch.zeros(4, 10, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        base(int_x, cat_x)
    base.apply(TU.init_weights)
    ckpt = {"model_state_dict": {k: v.detach().cpu().clone() for k, v in base.state_dict().items()}}
    del base
    models = []

    def probe(model, a, checkpoint):
        res = E.finetune_and_eval_one_model(model, a, checkpoint)
        torch.cuda.synchronize()
        models.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        return res

    def inline(self, choices, on_cpu, ckpt_holder, kwargs):
        out = []
        for job_id, ch in enumerate(choices):
            rd, holder = {}, {"ckpt": ckpt}
            a = argparse.Namespace(**vars(self._args))
            SU.create_model_train_and_get_results_helper(a, 0, self._eval_fn, self._tokenizer, ch, rd, holder, kwargs)
            out += [rd[k] for k in sorted(rd)]
        return out

    monkeypatch.setattr(S.Searcher, "_run_jobs", inline)
    np.random.seed(21)
    s = S.Searcher(probe, args)
    top = s.random_search_from_supernet(budget=3, top_k=2, num_parallel_workers=1, sorted=True)
    assert len(s.all_results) == 3 and len(top) == 2
    # the log lists hold one entry per test pass: exactly one with the recipe's --test_only_at_last_step 1
    assert all(len(r["test_loss"]) == 1 and np.isfinite(r["test_loss"][0]) and 0.0 <= r["test_auroc"][0] <= 1.0 for r in s.all_results)
    assert len({r["hash_token"] for r in s.all_results}) == 3
    assert top[0]["test_loss"][0] <= top[1]["test_loss"][0]
    for sd in models:
        moved = sorted(k for k in sd if not torch.equal(sd[k], ckpt["model_state_dict"][k]))
        assert moved == ["_final.bias", "_final.weight"], moved


def test_table_sharding_row_flag_trains_like_whole_tables(tmp_path, capsys):
    """`main_train.py --table-sharding row` (row-sharded tables as a usable mode) against the same command line with
    whole tables: identical seeds, so identical initial weights and batches — train / test losses of every logged step and the saved
    checkpoint (whole tables under the reference's keys, Adagrad sums included) must agree to fp32 rounding.  Single rank here: the
    route degenerates to the identity, every kernel and the whole step composition of the sharded mode runs; the N > 1 routing is the
    gloo world-2 test's (tests/test_sharded_tables_cpu.py)."""
    root = _shards(tmp_path)
    runs = []
    for mode in ("none", "row"):
        logdir = str(tmp_path / ("logs_" + mode))
        args = MT.build_parser().parse_args([
            "--root_dir", root, "--net", "supernet-config", "--supernet_config", CFG, "--num_epochs", "1", "--learning_rate", "0.1",
            "--train_batch_size", "8", "--test_batch_size", "16", "--wd", "0", "--logging_dir", logdir, "--gpu", "0", "--test_interval", "4",
            "--display_interval", "2", "--train_limit", "96", "--table-sharding", mode])
        torch.manual_seed(0)
        np.random.seed(0)
        logs = MT.main(args)
        ck = torch.load(os.path.join(logdir, "supernet-config_checkpoint.pt"), map_location="cpu")
        runs.append((logs[0], ck))
    capsys.readouterr()
    (la, ca), (lb, cb) = runs
    assert la["iters"] == lb["iters"]
    assert np.allclose(la["train_loss"], lb["train_loss"], rtol=1e-5, atol=1e-6), (la["train_loss"], lb["train_loss"])
    assert np.allclose(la["test_loss"], lb["test_loss"], rtol=1e-5, atol=1e-6)
    assert list(ca["model_state_dict"]) == list(cb["model_state_dict"])
    for k, v in ca["model_state_dict"].items():
        assert v.shape == cb["model_state_dict"][k].shape and torch.allclose(v, cb["model_state_dict"][k], rtol=0, atol=2e-5), k
    sa, sb = ca["optimizer_state_dict"]["state"], cb["optimizer_state_dict"]["state"]
    assert len(sa) == len(sb)
    for i in sa:
        assert torch.allclose(sa[i]["sum"], sb[i]["sum"], rtol=1e-4, atol=1e-7), i


def test_table_sharding_row_refuses_the_torch_route(tmp_path):
    """row-sharded tables have no differentiable forward: a recipe that needs the torch route (weight decay != 0) must fail loudly, not
    train on stale or partial tables"""
    from nasrec_amd._lib import EngineError
    root = _shards(tmp_path)
    args = MT.build_parser().parse_args([
        "--root_dir", root, "--net", "supernet-config", "--supernet_config", CFG, "--learning_rate", "0.1", "--train_batch_size", "8",
        "--test_batch_size", "16", "--wd", "1e-4", "--logging_dir", str(tmp_path / "l"), "--gpu", "0", "--train_limit", "48", "--table-sharding", "row"])
    with pytest.raises(EngineError):
        MT.main(args)


def test_train_supernet_cli_with_row_sharded_tables(tmp_path, capsys):
    args = TS.build_parser().parse_args([
        "--dataset", "kdd", "--root_dir", "synthetic:steps=6,test_steps=1,seed=3", "--logging_dir", str(tmp_path / "sn"), "--config", "autoctr",
        "--num_blocks", "3", "--use_layernorm", "1", "--strategy", "default", "--anypath_choice", "binomial-0.5", "--supernet_training_steps", "2",
        "--train_batch_size", "16", "--test_batch_size", "16", "--train_limit", "64", "--learning_rate", "0.05", "--display_interval", "2",
        "--gpu", "0", "--table-sharding", "row"])
    np.random.seed(0)
    torch.manual_seed(0)
    logs = TS.main(args)
    capsys.readouterr()
    assert len(logs[0]["test_loss"]) == 1 and np.isfinite(logs[0]["test_loss"][0]) and all(np.isfinite(v) for v in logs[0]["train_loss"])


def test_eval_subnet_from_scratch_trains_sampled_subnetworks(tmp_path, capsys):
    """nasrec/eval_subnet_from_scratch.py:195-245 on the engine: `--num_subnets` fixed sub-networks drawn by `fixed-path` from the seeded
    global stream, each initialised with init_weights, trained for an epoch (the fused step: Adagrad, weight decay 0) and tested once at
    its end; results.pickle holds one record per sub-network, distinct architectures, finite scores; a second run with the same seed
    draws the same architectures."""
    from nasrec_amd import eval_subnet_from_scratch as ES
    from nasrec_amd.utils.config import DATASETS
    from nasrec_amd.utils.io_utils import load_pickle_data
    tables = [min(n, 1000) for n in DATASETS["kdd"]["tables"]]

    def run(logdir):
        args = ES.build_parser().parse_args([
            "--dataset", "kdd", "--root_dir", "synthetic:steps=6,test_steps=2,seed=9,cap=1000", "--logging_dir", logdir, "--config", "autoctr",
            "--num_blocks", "3", "--use_layernorm", "0", "--num_subnets", "2", "--random_seed", "7", "--learning_rate", "0.05",
            "--train_batch_size", "32", "--test_batch_size", "32", "--train_limit", "192", "--gpu", "0"])
        args.num_embeddings = tables
        torch.manual_seed(1)
        res = ES.main(args)
        return res, load_pickle_data(os.path.join(logdir, "results.pickle"))

    res, saved = run(str(tmp_path / "a"))
    out = capsys.readouterr().out
    assert "Evaluating 0 out of 2 subnetworks!" in out and "Trained model with the following choice..." in out and "FLOPS:" in out
    assert len(res) == 2 and len(saved) == 2
    assert all(np.isfinite(r["test_loss"]) and 0.0 <= r["test_auroc"] <= 1.0 for r in res)
    assert json.dumps(res[0]["choice"], sort_keys=True, default=str) != json.dumps(res[1]["choice"], sort_keys=True, default=str)
    res2, _ = run(str(tmp_path / "b"))
    capsys.readouterr()
    assert [json.dumps(r["choice"], sort_keys=True, default=str) for r in res2] == [json.dumps(r["choice"], sort_keys=True, default=str) for r in res]
