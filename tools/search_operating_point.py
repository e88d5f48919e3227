#!/usr/bin/env python3
"""The search loop at ITS operating point: what one candidate of the regularised-evolution search costs
on one MI355X with the recipe of /root/reference scripts/run_ea/criteo_run_ea_from_supernet_xlarge.sh — Criteo xlarge weight-sharing
supernet (7 blocks, LayerNorm), full tables, the candidate's path pinned, `set_mode_to_finelune_last_only`, Adagrad(lr 0.04, eps 1e-2),
500 training steps at batch 512 and 150 evaluation batches of 8192, one test pass at the end — through the SAME functions the search
CLI runs (`searcher_utils._create_model_train_and_get_results` -> `eval_subnet_from_supernet.finetune_and_eval_one_model` ->
`train_and_test_one_epoch`), on the synthetic pipe (`--root_dir synthetic:...`: host-generated Criteo-shaped batches, H2D per batch).
The only timing the reference publishes for anything is a comment at this operating point (eval_subnet_from_supernet.py:114-115, Tesla M40):
0.05 - 0.06 s per batch-512 step with last-layer-only fine-tuning, 0.21 - 0.23 s fine-tuning the whole network.

Printed: per candidate the wall time of the whole call and of its phases (model build + full-path warm-up + checkpoint load; the 500
training steps; the 150 evaluation batches incl. AUROC on the host), candidates / hour / GPU; then the same three step kinds on batches
already resident in HBM (no pipe, no Python harness): last-layer-only step, whole-network fine-tune step, evaluation forward.

    python tools/search_operating_point.py [--candidates 3] [--train_steps 500] [--eval_steps 150]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=3)
    ap.add_argument("--train_steps", type=int, default=500)
    ap.add_argument("--eval_steps", type=int, default=150)
    ap.add_argument("--train_batch_size", type=int, default=512)
    ap.add_argument("--test_batch_size", type=int, default=8192)
    ap.add_argument("--cap", type=int, default=0, help="cap the tables (0 = the full 33.76 M rows)")
    ap.add_argument("--profile", action="store_true", help="cProfile of the LAST candidate's call (host time by function)")
    a = ap.parse_args()

    from nasrec_amd import eval_subnet_from_supernet as E
    from nasrec_amd.searcher import searcher as S
    from nasrec_amd.searcher import searcher_utils as SU
    from nasrec_amd.utils import train_utils as TU
    from nasrec_amd.utils.config import DATASETS

    tables = list(DATASETS["criteo"]["tables"])
    if a.cap:
        tables = [min(n, a.cap) for n in tables]
    root = "synthetic:steps=%d,test_steps=%d,seed=11" % (a.train_steps, a.eval_steps) + (",cap=%d" % a.cap if a.cap else "")
    args = E.build_parser().parse_args([
        "--dataset", "criteo-kaggle", "--root_dir", root, "--logging_dir", "/tmp/nasrec_search_point", "--config", "xlarge", "--num_blocks", "7",
        "--use_layernorm", "1", "--learning_rate", "0.04", "--wd", "0", "--max_train_steps", str(a.train_steps), "--max_eval_steps", str(a.eval_steps),
        "--train_batch_size", str(a.train_batch_size), "--test_batch_size", str(a.test_batch_size), "--method", "random", "--test_only_at_last_step", "1",
        "--finetune_whole_supernet", "0", "--display_interval", "1000000", "--gpu", "0"])
    args.num_embeddings = tables

    # the supernet checkpoint every candidate starts from (init_weights-scale weights: nothing here depends on their values)
    torch.manual_seed(3)
    np.random.seed(21)
    t0 = time.perf_counter()
    base = SU.build_supernet(args, tables).to(0)
    with torch.no_grad():
        base(torch.zeros(4, 13, device="cuda"), torch.zeros(4, 26, dtype=torch.int64, device="cuda"))
    base.apply(TU.init_weights)
    ckpt = {"model_state_dict": {k: v.detach().clone() for k, v in base.state_dict().items()}}  # (kept on the device: the reference shares it through a manager dict)
    torch.cuda.synchronize()
    print("checkpoint of the supernet built in %.1f s (%d parameter tensors, %.2f GB)" % (
        time.perf_counter() - t0, len(ckpt["model_state_dict"]), sum(v.numel() * 4 for v in ckpt["model_state_dict"].values()) / 1e9))
    del base
    torch.cuda.empty_cache()

    # phase timers: wrap the harness's own functions
    phase = {}
    real_train, real_test = E.train_and_test_one_epoch, TU.test_one_epoch

    def timed_train(*x, **kw):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = real_train(*x, **kw)
        torch.cuda.synchronize()
        phase["train_and_test"] = time.perf_counter() - t
        return r

    def timed_test(*x, **kw):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = real_test(*x, **kw)
        torch.cuda.synchronize()
        phase["test"] = phase.get("test", 0.0) + time.perf_counter() - t
        return r
    E.train_and_test_one_epoch = timed_train
    TU.test_one_epoch = timed_test

    searcher = S.Searcher(E.finetune_and_eval_one_model, args)
    tok = searcher._tokenizer
    rows = []
    for c in range(a.candidates):
        choice = None  # (random search: `configure_path_sampling_strategy("fixed-path")` draws the candidate, searcher.py:134-152)
        phase.clear()
        torch.cuda.synchronize()
        t = time.perf_counter()
        prof = None
        if a.profile and c == a.candidates - 1:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        res = SU._create_model_train_and_get_results(argparse.Namespace(**vars(args)), 0, E.finetune_and_eval_one_model, tok, choice, ckpt, {"beta": 0.0})
        if prof is not None:
            prof.disable()
            import io
            import pstats
            buf = io.StringIO()
            pstats.Stats(prof, stream=buf).sort_stats("cumulative").print_stats(45)
            print(buf.getvalue())
        torch.cuda.synchronize()
        total = time.perf_counter() - t
        tr = phase.get("train_and_test", 0.0) - phase.get("test", 0.0)
        rows.append((total, total - phase.get("train_and_test", 0.0), tr, phase.get("test", 0.0), res["test_loss"][-1], res["test_auroc"][-1]))
        print("candidate %d: %.2f s = set-up (build, full-path warm-up, checkpoint load) %.2f + %d training steps %.2f (%.3f ms/step, %.0f samples/s) "
              "+ %d evaluation batches %.2f (%.2f ms/batch, %.2f M samples/s);  test loss %.4f AUROC %.4f" % (
                  c, total, rows[-1][1], a.train_steps, tr, tr / a.train_steps * 1e3, a.train_batch_size * a.train_steps / max(tr, 1e-9),
                  a.eval_steps, rows[-1][3], rows[-1][3] / a.eval_steps * 1e3, a.test_batch_size * a.eval_steps / max(rows[-1][3], 1e-9) / 1e6,
                  rows[-1][4], rows[-1][5]))
    E.train_and_test_one_epoch, TU.test_one_epoch = real_train, real_test
    if rows:
        use = rows[1:] if len(rows) > 1 else rows  # (the first candidate also pays one-off costs: kernels' first launches, allocator growth)
        mean = float(np.mean([r[0] for r in use]))
        print("steady state: %.2f s per candidate = %.0f candidates / hour / GPU  (reference, Tesla M40, from its own comment: 500 x 0.05 - 0.06 s = 25 - 30 s of "
              "training steps alone per candidate)" % (mean, 3600.0 / mean))

    # ---- the same step kinds on batches resident in HBM: no pipe, no harness loop ----------------------------------------------
    import bench  # (synthetic_batches: the bench's input generator)
    model = SU.build_supernet(args, tables).to(0)
    with torch.no_grad():
        model(torch.zeros(4, 13, device="cuda"), torch.zeros(4, 26, dtype=torch.int64, device="cuda"))
    model.load_state_dict(ckpt["model_state_dict"], strict=True)
    model.configure_path_sampling_strategy("fixed-path")
    loss_fn = torch.nn.BCEWithLogitsLoss()
    tb = bench.synthetic_batches(8, a.train_batch_size, 13, tables, "cuda", 100)
    eb = bench.synthetic_batches(4, a.test_batch_size, 13, tables, "cuda", 200)

    def timeit(fn, n, w=10):
        for i in range(w):
            fn(i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            fn(w + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    model.set_mode_to_finelune_last_only()
    opt = torch.optim.Adagrad(model.parameters(), lr=0.04, eps=1e-2)

    def last_only(i):
        int_x, cat_x, y = tb[i % len(tb)]
        opt.zero_grad()
        loss_fn(model(int_x, cat_x), y.view(-1, 1)).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
    dt = timeit(last_only, 200)
    print("resident batches, last-layer-only fine-tune step (B = %d): %.3f ms = %.0f samples/s   (reference comment: 50 - 60 ms on a Tesla M40)" % (
        a.train_batch_size, dt * 1e3, a.train_batch_size / dt))

    def evaluate(i):
        with torch.no_grad():
            model(eb[i % len(eb)][0], eb[i % len(eb)][1])
    model.eval()
    dt = timeit(evaluate, 100)
    print("resident batches, evaluation forward (B = %d): %.3f ms = %.2f M samples/s" % (a.test_batch_size, dt * 1e3, a.test_batch_size / dt / 1e6))
    model.train()
    model.set_mode_to_normal_mode()
    lr = 0.04

    def whole(i):
        int_x, cat_x, y = tb[i % len(tb)]
        model.engine_train_step(int_x, cat_x, y.view(-1), lr)
    dt = timeit(whole, 200)
    print("resident batches, whole-network fine-tune step, fused engine step (B = %d): %.3f ms = %.0f samples/s   (reference comment: 210 - 230 ms on a Tesla M40)" % (
        a.train_batch_size, dt * 1e3, a.train_batch_size / dt))


if __name__ == "__main__":
    main()
