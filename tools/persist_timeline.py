#!/usr/bin/env python3
"""Timeline of the persistent step (NASREC_OP_PERSIST) of the cfg-2 plan: every workgroup stamps the 100 MHz wall clock at entry, with
its dependencies satisfied, behind its body and behind its arrival (nasrec_persist_desc_t.trace); printed per item: when its first
workgroup entered, when its LAST dependency-wait ended, when its last body ended, and — per dependency edge on the path that ends last —
how long after the producer's last body the consumer's wait ended (the seam).  `python tools/persist_timeline.py [steps]`"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nasrec_amd import _lib as L, plan as P
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.utils.config import DATASETS

ds = DATASETS["criteo"]
tables, Fd, Fs = ds["tables"], ds["Fd"], ds["Fs"]
ca = json.load(open(os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
choice = {"macro": ca["macro"], "micro": ca["micro"]}
cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
eng = SupernetEngine(cfg, Fd, Fs, tables, warm_choice=choice)
eng.init_weights(seed=0)
eng.persist = True
batches = bench.synthetic_batches(8, 256, Fd, tables, "cuda", 1)
for i in range(20):
    eng.train_step(*batches[i % 8], 1e-3, choice, graph=False)
torch.cuda.synchronize()
cp = eng._last_plan[2]
pds = [d for d in cp.fb.descs if isinstance(d, L.PersistDesc)]
print("fb program:", [type(d).__name__ for d in cp.fb.descs])
names = {getattr(L, k): k[3:] for k in dir(L) if k.startswith("OP_")}
for d in pds:
    tr = torch.zeros(d.total_blocks * 4, dtype=torch.int64, device="cuda")
    d.trace = tr.data_ptr()
    for i in range(5):
        eng.train_step(*batches[i % 8], 1e-3, choice, graph=False)
    torch.cuda.synchronize()
    d.trace = None
    t = tr.cpu().numpy().reshape(-1, 4).astype(np.float64) / 100.0  # us
    t0 = t[:, 0].min()
    t -= t0
    import ctypes as C
    items = (L.PersistItem * d.n).from_address(d.host_items)
    # host_items are not updated by a device prepare: recompute first/nwg from a dry prepare of a copy
    dd = L.PersistDesc.from_buffer_copy(d)
    dd.items = None
    L.check(L.load().nasrec_persist_prepare(C.addressof(dd)))
    rows = []
    for k in range(d.n):
        it = items[k]
        nwg = it._pad[1]
        sl = t[it.first:it.first + nwg]
        rows.append(dict(k=k, kind=names.get(it.kind, "?") + ({0: "", 1: ":main", 2: ":epi"}[it.part] if it.kind == L.OP_GEMM else ""), nwg=nwg, nblk=it.nblk,
                         enter=sl[:, 0].min(), enter_last=sl[:, 0].max(), ready=sl[:, 1].max(), body_end=sl[:, 2].max(), done=sl[:, 3].max(),
                         deps=[it.deps[q] for q in range(it.ndeps)]))
    if os.environ.get("NASREC_TIMELINE_JSON"):
        json.dump([dict(r, node=d.nodes[r["k"]].index) for r in rows], open(os.environ["NASREC_TIMELINE_JSON"] + ".%d" % pds.index(d), "w"))
    print("persistent launch: %d items, %d workgroups, %.1f us from the first entry to the last arrival" % (d.n, d.total_blocks, t[:, 3].max()))
    print("  k kind              wgs units  enter(first..last)   ready   body_end   done | per dependency: producer done -> this ready (seam)")
    for r in rows:
        seams = ["%d:%+.1f" % (j, r["ready"] - rows[j]["done"]) for j in r["deps"]]
        print("%3d %-16s %5d %5d  %7.1f .. %7.1f  %7.1f  %8.1f %7.1f | %s" % (r["k"], r["kind"], r["nwg"], r["nblk"], r["enter"], r["enter_last"], r["ready"], r["body_end"], r["done"], " ".join(seams)))
    # the chain that ends last
    k = max(range(d.n), key=lambda i: rows[i]["done"])
    chain = []
    while True:
        chain.append(k)
        if not rows[k]["deps"]:
            break
        k = max(rows[k]["deps"], key=lambda j: rows[j]["done"])
    chain.reverse()
    print("  last-finishing chain:", " -> ".join("%d(%s %.1f..%.1f)" % (k, rows[k]["kind"], rows[k]["ready"], rows[k]["done"]) for k in chain))
