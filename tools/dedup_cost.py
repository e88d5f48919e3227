#!/usr/bin/env python3
"""Cost of the GLOBAL-batch row-gradient dedup + row Adagrad that every rank of a data-parallel step runs (replicated tables: each rank
applies the gathered row gradients of all ranks, nasrec_amd/parallel.py): cfg 2 at 8 / 4 / 2 GPUs = 2048 / 1024 / 512 samples x 26 fields, cfg 5 at 8 GPUs = 65 536 samples x 10 fields, cfg 3 / 4 at
8 GPUs = 32 768 x 26 / 23.  One GPU is enough to time it: the kernels do not care where the rows came from."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nasrec_amd import _lib as L, plan as P
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.utils.config import DATASETS

lib = L.load()
for cfgid, world in ((2, 8), (2, 4), (2, 2), (2, 1), (5, 8), (3, 8), (4, 8), (5, 1)):
    w = bench.WORKLOADS[cfgid]
    ds = DATASETS[w["dataset"]]
    tables = [min(n, w["cap"]) if w.get("cap") else n for n in ds["tables"]]
    Bg, Fs = w["B"] * world, ds["Fs"]
    cfg = P.NetConfig(1, ops_config_lib["autoctr"], True, "relu", fixed=False)
    eng = SupernetEngine(cfg, ds["Fd"], Fs, tables)
    eng._ensure_table_state()
    g = torch.Generator().manual_seed(1)
    cat = bench.synthetic_ids(Bg, tables, g).cuda()
    sg = torch.randn(Bg * Fs * 16, device="cuda") * 1e-3

    class H:
        pass
    import nasrec_amd.engine as E
    sp = torch.cuda.current_stream().cuda_stream
    names = {getattr(L, n): n[3:] for n in dir(L) if n.startswith("OP_")}
    parts = []
    for label, cap in (("two halves", E.DEDUP_SPLIT_MAX_B), ("one launch", 0)):
        if label == "two halves" and Bg > L.DEDUP_IDS_MAX_B:
            continue
        keep = E.DEDUP_SPLIT_MAX_B
        E.DEDUP_SPLIT_MAX_B = cap
        h = H()
        sgc = sg.clone()  # (the two-halves launch sums in place)
        descs = eng._optimizer_descs(h, Bg, cat, sgc, 5.0, 1e-2)
        E.DEDUP_SPLIT_MAX_B = keep
        row = []
        if h.dedup_ids is not None:  # off the tail: beside the forward (N > 1) / on the staging launch (B <= 256)
            row.append("[hidden: DEDUP_IDS %.1f us]" % (bench.time_desc(lib, L, sp, h.dedup_ids, iters=20) * 1e3))
        tail = 0.0
        for d in descs:
            us = bench.time_desc(lib, L, sp, d, iters=20) * 1e3
            tail += us
            row.append("%s %.1f us" % (names[d.kind], us))
        parts.append("%s: %s = %.1f us behind the backward" % (label, ", ".join(row), tail))
    torch.cuda.synchronize()
    eng.check_indices()
    print("cfg %d x %d GPUs: global batch %d x %d fields (%.1f MB of row gradients)\n    %s" % (cfgid, world, Bg, Fs, Bg * Fs * 64 / 1e6, "\n    ".join(parts)))
    del eng, cat, sg
    torch.cuda.empty_cache()
