"""Input pipeline vs golden vectors produced by the REAL reference readers/transforms (tests/golden/make_golden_datapipes.py):
every batch — values, dtypes, shapes and arrival order over two shards of different length — must be identical."""
import argparse
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from nasrec_amd.utils import data_pipes as DP


@pytest.mark.parametrize("reader", ["native", "python"])
@pytest.mark.parametrize("ds", ["criteo-kaggle", "avazu", "kdd"])
def test_tsv_pipeline_matches_reference_batches(ds, reader, tmp_path, monkeypatch):
    monkeypatch.setenv("NASREC_TSV_READER", reader)
    z = np.load(os.path.join(GOLDEN, "datapipes.npz"), allow_pickle=False)
    for s in range(2):
        d = tmp_path / ("shard-%d" % s)
        d.mkdir()
        for name in ("trainval.txt", "test.txt"):
            (d / name).write_text(str(z["%s/shard-%d/%s" % (ds, s, name)]) + "\n")
    args = argparse.Namespace(dataset=ds, root_dir=str(tmp_path), train_split="trainval", validate_split="test", train_batch_size=8,
                              test_batch_size=16)
    train, test = DP.make_loaders(args)
    assert type(train.pipes[0]) is (DP.NativeTsvShard if reader == "native" else DP.TsvShard)
    for split, loader in (("train", train), ("test", test)):
        got = list(loader)
        assert len(got) == int(z["%s/%s/n" % (ds, split)])
        for i, (int_x, cat_x, y) in enumerate(got):
            assert int_x.dtype == torch.float32 and cat_x.dtype == torch.int64 and y.dtype == torch.float32
            # log() of an integer tensor: identical on the CPU that produced the fixture, within 1-2 ulp on another
            # micro-architecture (torch's vectorised logf differs between AVX2 and AVX-512 builds)
            assert np.allclose(int_x.numpy(), z["%s/%s/%d/int_x" % (ds, split, i)], rtol=3e-7, atol=0)
            assert np.array_equal(cat_x.numpy(), z["%s/%s/%d/cat_x" % (ds, split, i)])  # bit-exact
            assert np.array_equal(y.numpy(), z["%s/%s/%d/y" % (ds, split, i)])
        assert len(list(loader)) == len(got)  # re-iterable (one pass per epoch)


def test_missing_and_out_of_vocabulary_ids():
    spec = DP.DatasetSpec("toy", 1, 2, [5, 1000])
    rows = [["1", "-3", "", "ffffffff"], ["0", "junk", "4", "3e8"]]
    int_x, cat_x, y = DP.rows_to_batch(rows, spec)
    assert cat_x.tolist() == [[0, 0xffffffff % 999 + 1], [4 % 4 + 1, 1000 % 999 + 1]]
    assert int_x.view(-1).tolist() == [0.0, 0.0] and y.view(-1).tolist() == [1.0, 0.0]
    with pytest.raises(ValueError):
        DP.rows_to_batch([["1", "2"]], spec)


def test_synthetic_source_shapes():
    args = argparse.Namespace(dataset="avazu", root_dir="synthetic:steps=3,test_steps=2,seed=7", train_split="train", validate_split="test",
                              train_batch_size=4, test_batch_size=6)
    train, test = DP.make_loaders(args)
    b = list(train)
    assert len(b) == 3 and b[0][0].shape == (4, 1) and b[0][1].shape == (4, 23) and b[0][2].shape == (4, 1)
    assert float(b[0][0].abs().sum()) == 0.0 and len(list(test)) == 2
    assert all(int(b[0][1][:, f].max()) < n for f, n in enumerate(DP.SPECS["avazu"].tables))


def test_native_reader_hands_python_only_constructs_back_and_agrees(tmp_path):
    """quotes (csv semantics), whitespace / underscores inside integers, hex prefixes, CRLF line ends, a missing final
    newline and tiny read chunks: the native reader must return exactly what the pure-Python reader returns"""
    spec = DP.DatasetSpec("toy", 2, 3, [7, 100, 65536])
    lines = ["1\t5\t-2\tff\t\t1f", "0\t 7 \t1_0\t0x1F\tabc\tABC", '1\t"3"\t4\t"a"\tb\tc', "0\t\tx\t\t\t", "1\t+9\t--1\t7fffffff\t0\tdeadbeef",
             "1\t3\t4\t00000000000000ff\t5\t6"]
    for i in range(40):
        lines.append("%d\t%d\t%d\t%x\t%x\t%x" % (i & 1, i * 37, -i, i * 7919, i, i * i * 104729))
    p = tmp_path / "x.txt"
    p.write_bytes("\r\n".join(lines).encode())  # CRLF, no newline at the end
    want = list(DP.TsvShard(str(p), spec, 8))
    for chunk in (13, 64, 1 << 20):
        nat = DP.NativeTsvShard(str(p), spec, 8)
        nat.chunk_bytes = chunk
        got = list(nat)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            for x, y in zip(a, b):
                assert x.dtype == y.dtype and torch.equal(x, y)
    bad = tmp_path / "bad.txt"
    bad.write_text("1\t2\t3\n")
    with pytest.raises(ValueError):
        list(DP.NativeTsvShard(str(bad), spec, 8))


def test_abandoned_iterators_stop_their_reader_threads(tmp_path, monkeypatch):
    """`next(iter(loader))`, `break` on max_train_steps: the harness abandons loader iterators all the time (train_utils.py:152,
    the warm-ups, eval_subnet_from_supernet for every candidate).  A reader thread must not outlive its iterator blocked in
    `put` (with device staging it would hold GPU blocks and pinned memory), and `peek()` returns the same first batch without
    starting one."""
    import gc
    import threading
    import time
    monkeypatch.setenv("NASREC_TSV_READER", "native")
    z = np.load(os.path.join(GOLDEN, "datapipes.npz"), allow_pickle=False)
    for s in range(2):
        d = tmp_path / ("shard-%d" % s)
        d.mkdir()
        body = str(z["criteo-kaggle/shard-%d/trainval.txt" % s]) + "\n"
        (d / "trainval.txt").write_text(body * 40)  # long enough that a reader fills its queue and blocks
        (d / "test.txt").write_text(body)
    args = argparse.Namespace(dataset="criteo-kaggle", root_dir=str(tmp_path), train_split="trainval", validate_split="test",
                              train_batch_size=2, test_batch_size=2)
    train, _ = DP.make_loaders(args)
    assert train.prefetch
    base = threading.active_count()
    first = next(iter(train))
    for _ in range(5):
        b = next(iter(train))
        assert all(torch.equal(x, y) for x, y in zip(b, first))
    it = iter(train)
    for k, _ in enumerate(it):
        if k == 3:
            break
    del it
    gc.collect()
    deadline = time.time() + 5.0
    while threading.active_count() > base and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() == base, "reader threads of abandoned iterators are still alive"
    p = train.peek()
    assert threading.active_count() == base and all(torch.equal(x, y) for x, y in zip(p, first))
    assert len(list(train)) == len(list(train))  # a complete pass still works, twice
