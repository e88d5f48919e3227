#!/usr/bin/env python3
"""Sweep the split-K factor of the GEMM launches of the bench step (one at a time, timed in isolation with HIP events)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nasrec_amd import _lib as L, plan as P
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.utils.config import NUM_EMBEDDINGS_CRITEO

lib = L.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ca = json.load(open(os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
choice = {"macro": ca["macro"], "micro": ca["micro"]}
cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
eng = SupernetEngine(cfg, 13, 26, NUM_EMBEDDINGS_CRITEO, device=dev, warm_choice=choice)
eng.init_weights(seed=0)
cp = eng.compile(choice, 256, train=True)
sp = eng.stream.cuda_stream
ws = torch.zeros(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)  # 64 MB scratch for any split
with torch.cuda.stream(eng.stream):
    for phase, prog in (("fwd", cp.fwd), ("bwd", cp.bwd)):
        for d in prog.descs:
            if not isinstance(d, L.GemmDesc):
                continue
            segs = [(d.seg[q].M, d.seg[q].N, d.seg[q].K) for q in range(d.nseg) if d.seg[q].A]
            if not segs:
                continue
            Kt = max(s[2] for s in segs) if d.zmode else sum(s[2] for s in segs)
            if Kt < 256:
                continue
            old_s, old_ws = d.splitk, d.workspace
            res = []
            for S in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32):
                M, N = max(s[0] for s in segs), max(s[1] for s in segs)
                nprob = len(segs) if d.zmode else 1
                if S * M * N * nprob * 4 > ws.numel() * 4 or S > (Kt + 31) // 32:
                    continue
                d.splitk, d.workspace = S, ws.data_ptr()
                res.append((S, bench.time_desc(lib, L, sp, d, iters=60) * 1e3))
            d.splitk, d.workspace = old_s, old_ws
            best = min(res, key=lambda r: r[1])
            cur = [r for r in res if r[0] == max(1, old_s)]
            print("%s am=%d bm=%d z=%d M=%d N=%d K=%s  plan S=%d (%.1f us)  best S=%d (%.1f us)  | %s" % (
                phase, d.amode, d.bmode, d.zmode, segs[0][0], segs[0][1], "+".join(str(s[2]) for s in segs) if not d.zmode else "%dx%d" % (len(segs), Kt),
                old_s, cur[0][1] if cur else -1, best[0], best[1], " ".join("%d:%.1f" % r for r in res)))
