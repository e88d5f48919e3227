#!/usr/bin/env python3
"""Throughput of the weight-sharing supernet step (BASELINE.json configs 3-5) through the drop-in SuperNet module:
path sampled per step from the global np.random stream, engine_train_step (forward, BCE, backward, clip, Adagrad).

    python tools/bench_supernet.py --dataset criteo --space xlarge --batch 4096 --steps 30
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nasrec_amd.search_space import ops_config_lib  # noqa: E402
from nasrec_amd.supernet.supernet import SuperNet  # noqa: E402
from nasrec_amd.utils.config import DATASETS, MAX_NUM_EMBEDDINGS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="criteo")
    ap.add_argument("--space", default="xlarge")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--strategy", default="default")
    ap.add_argument("--anypath", default="binomial-0.5")
    ap.add_argument("--cap", type=int, default=MAX_NUM_EMBEDDINGS)
    ap.add_argument("--blocks", type=int, default=7)
    args = ap.parse_args()
    ds = DATASETS[args.dataset]
    tables = [min(n, args.cap) for n in ds["tables"]]
    B = args.batch
    torch.manual_seed(0)
    np.random.seed(0)
    t0 = time.perf_counter()
    model = SuperNet(num_blocks=args.blocks, ops_config=ops_config_lib[args.space], use_layernorm=True, num_embeddings=tables,
                     sparse_input_size=ds["Fs"], path_sampling_strategy="full-path", fixed=False, anypath_choice=args.anypath).to("cuda")
    g = torch.Generator().manual_seed(1234)
    int_x = (torch.zeros(B, ds["Fd"]) if args.dataset == "avazu" else torch.log(torch.randint(0, 1000, (B, ds["Fd"]), generator=g).float() + 1)).cuda()
    cat_x = torch.stack([torch.randint(0, n, (B,), generator=g) for n in tables], 1).cuda()
    y = (torch.rand(B, generator=g) < 0.25).float().cuda()
    with torch.no_grad():
        model(int_x, cat_x)  # full-path warm-up (train_utils.py:413-433)
    model._engine.init_weights(0)
    model.configure_path_sampling_strategy(args.strategy)
    torch.cuda.synchronize()
    print("build + warm-up: %.1f s; dense params %d" % (time.perf_counter() - t0, model._engine.flat_numel))
    times = []
    for i in range(args.warmup + args.steps):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss = model.engine_train_step(int_x, cat_x, y, lr=1e-3)
        torch.cuda.synchronize()
        if i >= args.warmup:
            times.append(time.perf_counter() - t1)
    times = np.array(times)
    print("loss %.4f  median %.2f ms/step  mean %.2f  min %.2f  max %.2f  -> %.0f samples/s (median)" % (
        float(loss.item()), np.median(times) * 1e3, times.mean() * 1e3, times.min() * 1e3, times.max() * 1e3, B / np.median(times)))


if __name__ == "__main__":
    main()
