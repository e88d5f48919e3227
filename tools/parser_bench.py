#!/usr/bin/env python3
"""Rows/s of the input pipeline on synthetic Criteo-shaped TSV shards: the C parser alone (one thread, N threads) and the batches that
come out of the threaded NativeTsvShard readers (what main_train consumes)."""
import ctypes as C, os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nasrec_amd import _lib as L
from nasrec_amd.utils import data_pipes as D

rows = int(os.environ.get("ROWS", "400000"))
nshards = int(os.environ.get("SHARDS", "8"))
spec = D.SPECS["criteo-kaggle"]
rng = np.random.default_rng(0)
tmp = tempfile.mkdtemp(prefix="nasrec_tsv_")
lines = []
for r in range(rows):
    dense = ["%d" % v if v >= 0 else "" for v in rng.integers(-1, 5000, spec.Fd)]
    cat = ["%08x" % v if rng.random() > 0.03 else "" for v in rng.integers(0, 1 << 32, spec.Fs)]
    lines.append("\t".join(["%d" % rng.integers(0, 2)] + dense + cat))
blob = ("\n".join(lines) + "\n").encode()
paths = []
for s in range(nshards):
    d = os.path.join(tmp, "shard-%d" % s)
    os.makedirs(d)
    p = os.path.join(d, "train.txt")
    open(p, "wb").write(blob)
    paths.append(p)
print("%d rows per shard, %.1f MB, %.0f bytes per row, %d shards" % (rows, len(blob) / 1e6, len(blob) / rows, nshards))
lib = L.load()
tables = np.asarray(spec.tables, dtype=np.int64)


def parse_all(buf):
    lab, den, cat = np.empty(rows + 8, np.int64), np.empty((rows + 8, spec.Fd), np.int64), np.empty((rows + 8, spec.Fs), np.int64)
    consumed, status = C.c_int64(), C.c_int32()
    base = C.cast(C.c_char_p(buf), C.c_void_p).value
    return lib.nasrec_tsv_parse(base, len(buf), spec.Fd, spec.Fs, tables.ctypes.data, rows + 8, lab.ctypes.data, den.ctypes.data, cat.ctypes.data,
                                C.byref(consumed), C.byref(status))


t0 = time.perf_counter()
n = parse_all(blob)
dt = time.perf_counter() - t0
print("C parser, 1 thread: %.2f M rows/s (%.0f MB/s)" % (n / dt / 1e6, len(blob) / dt / 1e6))
for nt in (4, 8):
    ths = [threading.Thread(target=parse_all, args=(blob,)) for _ in range(nt)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print("C parser, %d threads: %.2f M rows/s aggregate" % (nt, nt * rows / dt / 1e6))
for bs in (256, 4096):
    pipes = [D.NativeTsvShard(p, spec, bs) for p in paths]
    loader = D.RoundRobinLoader(pipes, prefetch=True)
    t0 = time.perf_counter()
    k = 0
    for int_x, cat_x, y in loader:
        k += len(y)
    dt = time.perf_counter() - t0
    print("NativeTsvShard x %d threads, batch %d: %.2f M rows/s delivered as (int_x, cat_x, y) batches" % (nshards, bs, k / dt / 1e6))
import shutil
shutil.rmtree(tmp)
