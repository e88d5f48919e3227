"""Input pipeline vs golden vectors produced by the REAL reference readers/transforms (tests/golden/make_golden_datapipes.py):
every batch — values, dtypes, shapes and arrival order over two shards of different length — must be identical."""
import argparse
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from nasrec_amd.utils import data_pipes as DP


@pytest.mark.parametrize("ds", ["criteo-kaggle", "avazu", "kdd"])
def test_tsv_pipeline_matches_reference_batches(ds, tmp_path):
    z = np.load(os.path.join(GOLDEN, "datapipes.npz"), allow_pickle=False)
    for s in range(2):
        d = tmp_path / ("shard-%d" % s)
        d.mkdir()
        for name in ("trainval.txt", "test.txt"):
            (d / name).write_text(str(z["%s/shard-%d/%s" % (ds, s, name)]) + "\n")
    args = argparse.Namespace(dataset=ds, root_dir=str(tmp_path), train_split="trainval", validate_split="test", train_batch_size=8,
                              test_batch_size=16)
    train, test = DP.make_loaders(args)
    for split, loader in (("train", train), ("test", test)):
        got = list(loader)
        assert len(got) == int(z["%s/%s/n" % (ds, split)])
        for i, (int_x, cat_x, y) in enumerate(got):
            assert int_x.dtype == torch.float32 and cat_x.dtype == torch.int64 and y.dtype == torch.float32
            assert np.array_equal(int_x.numpy(), z["%s/%s/%d/int_x" % (ds, split, i)])  # bit-exact
            assert np.array_equal(cat_x.numpy(), z["%s/%s/%d/cat_x" % (ds, split, i)])
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
