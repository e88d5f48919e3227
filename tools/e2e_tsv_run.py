#!/usr/bin/env python3
"""End-to-end run of the harness on generated Criteo-format TSV shards: writes `rows` random rows over `shards` shard
directories under a temp dir, then runs nasrec_amd/main_train.py (best-1shot recipe, batch 256) for one epoch and prints
the wall-clock rate of the whole pipeline (native TSV reader threads -> H2D -> fused engine step)."""
import argparse, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--shards", type=int, default=4)
ap.add_argument("--prof", action="store_true", help="cProfile the epoch call: where the host's per-step time goes")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="nasrec_tsv_")
rng = np.random.default_rng(0)
t0 = time.time()
per = a.rows // a.shards
for s in range(a.shards):
    d = os.path.join(tmp, "shard-%d" % s)
    os.makedirs(d)
    ints = rng.integers(0, 1000, (per, 13))
    cats = rng.integers(0, 2 ** 32, (per, 26))
    lab = (rng.random(per) < 0.25).astype(np.int64)
    lines = ["%d\t%s\t%s" % (lab[i], "\t".join(map(str, ints[i])), "\t".join("%08x" % v for v in cats[i])) for i in range(per)]
    body = "\n".join(lines) + "\n"
    for name in ("trainval.txt",):
        open(os.path.join(d, name), "w").write(body)
    open(os.path.join(d, "test.txt"), "w").write("\n".join(lines[:4096]) + "\n")
print("generated %d rows in %.1f s under %s" % (a.rows, time.time() - t0, tmp))
from nasrec_amd import main_train as MT
args = MT.build_parser().parse_args([
    "--root_dir", tmp, "--net", "supernet-config",
    "--supernet_config", os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json"),
    "--learning_rate", "0.16", "--train_batch_size", "256", "--test_batch_size", "4096", "--wd", "0", "--logging_dir", os.path.join(tmp, "log"),
    "--gpu", "0", "--train_limit", str(a.rows), "--display_interval", "1000", "--test_interval", "1000000"])
t1 = time.time()
# time the epoch call on its own (model build and table initialisation are not pipeline time)
_orig = MT.train_and_test_one_epoch
epoch_s = []


def timed(*a_, **k_):
    import torch as _t
    _t.cuda.synchronize()
    t = time.time()
    if a.prof:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        r = _orig(*a_, **k_)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    else:
        r = _orig(*a_, **k_)
    _t.cuda.synchronize()
    epoch_s.append(time.time() - t)
    return r


MT.train_and_test_one_epoch = timed
logs = MT.main(args)
import torch
torch.cuda.synchronize()
dt = time.time() - t1
print("END-TO-END: %d rows, %.1f s wall (incl. model build, 2 test passes) -> %.0f rows/s; the epoch call alone (reader threads -> H2D -> fused step, "
      "incl. its test passes over 4096 rows x %d shards): %.2f s -> %.0f rows/s; final train loss %.4f" % (
    a.rows, dt, a.rows / dt, a.shards, epoch_s[0], a.rows / epoch_s[0], logs[0]["train_loss"][-1]))
