#!/usr/bin/env python3
"""Per-launch table of one training step of the bench workload (Criteo best-1shot, B=256): every descriptor of the step
timed on its own with HIP events (back-to-back launches of the same descriptor: L2 cold at every kernel boundary, MALL
warm — the conditions it sees inside the step)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nasrec_amd import _lib as L, plan as P, schedule as S
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.utils.config import NUM_EMBEDDINGS_CRITEO

B = int(os.environ.get("B", "256"))
ITEMS = float(os.environ.get("ITEMS", "0"))  # ITEMS=12: break worklists of >= 12 us down into their items
lib = L.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ca = json.load(open(os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
choice = {"macro": ca["macro"], "micro": ca["micro"]}
cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
eng = SupernetEngine(cfg, 13, 26, NUM_EMBEDDINGS_CRITEO, device=dev, warm_choice=choice)
eng.init_weights(seed=0)
bx = bench.synthetic_batches(1, B, 13, NUM_EMBEDDINGS_CRITEO, dev, 1)[0]
eng.train_step(bx[0], bx[1], bx[2], 1e-3, choice=choice)
cp = eng.compile(choice, B, train=True)
sp = eng.stream.cuda_stream
names = {getattr(L, n): n[3:] for n in dir(L) if n.startswith("OP_")}
tot = 0.0
rows = []
with torch.cuda.stream(eng.stream):
    # FB=1: the program the step really runs (forward and backward scheduled together, cp.fb) instead of the two separately packed ones
    progs = (("fb", cp.fb), ("opt", cp.opt)) if (os.environ.get("FB") == "1" and getattr(cp, "fb", None) is not None) else (("fwd", cp.fwd), ("bwd", cp.bwd), ("opt", cp.opt))
    for phase, prog in progs:
        for d in prog.descs:
            us = bench.time_desc(lib, L, sp, d, iters=100) * 1e3
            tot += us
            info = ""
            if isinstance(d, L.GemmDesc):
                segs = [(d.seg[q].M, d.seg[q].N, d.seg[q].K) for q in range(d.nseg) if d.seg[q].A]
                M, N = segs[0][0], segs[0][1]
                Kt = sum(s[2] for s in segs)
                fl = bench.gemm_flops(d)
                info = "am=%d bm=%d cm=%d z=%d M=%d N=%d K=%s splitk=%d act=%d  %.1f TF/s" % (
                    d.amode, d.bmode, d.cmode, d.zmode, M, N, "+".join(str(s[2]) for s in segs) if not d.zmode else "%dx%d" % (len(segs), segs[0][2]),
                    d.splitk, d.act, fl / us / 1e6)
            if isinstance(d, L.WorklistDesc):
                info = "worklist of " + " || ".join(names.get(n.desc.kind, "?") + ("[%s]" % n.part if n.part != "whole" else "") for n in d.nodes)
                if os.environ.get("MODEL") == "1":  # the balancing pass's estimate of this level (schedule._level_ns) beside the measurement
                    info = "[model %.1f us] " % (S._level_ns(d.nodes) / 1e3) + info
            rows.append((phase, names.get(d.kind, str(d.kind)), us, info))
            if isinstance(d, L.WorklistDesc) and ITEMS and us >= ITEMS:
                # every item as a one-item worklist of its own (same body, same descriptor): what it costs without the others
                for n in d.nodes:
                    b = S.item_bytes(n)
                    one = L.WorklistDesc()
                    one.kind, one.n = L.OP_WORKLIST, 1
                    it = one.item[0]
                    it.kind, it.part, it.off = n.desc.kind, S._PART[n.part], 0
                    C.memmove(C.addressof(one) + L.WorklistDesc.blob.offset, b, len(b))
                    ius = bench.time_desc(lib, L, sp, one, iters=50) * 1e3
                    dd = n.desc
                    what = names.get(dd.kind, "?") + ("[%s]" % n.part if n.part != "whole" else "")
                    if isinstance(dd, L.GemmDesc):
                        segs = [(dd.seg[q].M, dd.seg[q].N, dd.seg[q].K) for q in range(dd.nseg) if dd.seg[q].A]
                        what += " am=%d bm=%d cm=%d z=%d S=%d %s" % (dd.amode, dd.bmode, dd.cmode, dd.zmode, dd.splitk, segs[:4])
                        if os.environ.get("MODEL") == "1":  # (operand strides / accumulation / pointer alignment of the first live problem)
                            s0 = next(dd.seg[q] for q in range(dd.nseg) if dd.seg[q].A)
                            what += " ld=%d/%d/%d acc=%s act=%d A%%256=%d B%%256=%d C%%256=%d" % (s0.lda, s0.ldb, s0.ldc, [int(dd.seg[q].accumulate) for q in range(dd.nseg)] if dd.zmode else int(dd.beta != 0), dd.act, s0.A % 256, s0.B % 256, s0.C % 256)
                        if n.part != "epi":
                            what += "   [stand-alone kernel%s: %.2f us]" % (" + second pass" if dd.splitk > 1 else "", bench.time_desc(lib, L, sp, dd, iters=50) * 1e3)
                    if os.environ.get("MODEL") == "1":
                        what = "[model %.1f us, work %.1f] " % (S._BAL_COST(n) / 1e3, S._work_ns(n) / 1e3) + what
                    rows.append(("   ", "  item", ius, what))
for r in rows:
    print("%s %-14s %7.2f us  %s" % r)
if getattr(cp, "fb", None) is not None:  # the program the step actually runs: forward and backward scheduled together
    def short(d):
        if isinstance(d, L.WorklistDesc):
            return "WL[" + ",".join(names.get(n.desc.kind, "?")[:8] + (":" + n.part[0] if n.part != "whole" else "") for n in d.nodes) + "]"
        return names.get(d.kind, str(d.kind))
    print("joint forward+backward program (%d launches): %s" % (len(cp.fb.descs), "  ".join(short(d) for d in cp.fb.descs)))
print("sum of isolated launches: %.1f us over %d launches" % (tot, sum(1 for r in rows if r[1] != "  item")))

# host enqueue time vs GPU time of the graph-replayed step
import time
for _ in range(30):
    eng.train_step(bx[0], bx[1], bx[2], 1e-3, choice=choice, graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    eng.train_step(bx[0], bx[1], bx[2], 1e-3, choice=choice, graph=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("graph step: host enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
