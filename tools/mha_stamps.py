#!/usr/bin/env python3
"""In-kernel stage times of the Transformer backward body inside the batch-256 step (csrc/attention_body.h built with -DMHA_STAMPS:
tools/build_variant.sh mhastamps -DMHA_STAMPS; NASREC_HIP_LIB=nasrec_amd/lib/variants/mhastamps.so).  Wave 0 of every sample's
workgroup overwrites the first 13 words of its parameter-gradient partial with the shader clock (~2.2 GHz) at the stage boundaries (a timing-only
build: the gradients are wrong).  Printed: medians over the samples of the stage durations, in units of 100 clocks."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nasrec_amd import _lib as L, plan as P, schedule as S
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.utils.config import NUM_EMBEDDINGS_CRITEO


class _Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


B = int(os.environ.get("B", "256"))
lib = L.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ca = json.load(open(os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
choice = {"macro": ca["macro"], "micro": ca["micro"]}
cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
eng = SupernetEngine(cfg, 13, 26, NUM_EMBEDDINGS_CRITEO, device=dev, warm_choice=choice)
eng.init_weights(seed=0)
bx = bench.synthetic_batches(1, B, 13, NUM_EMBEDDINGS_CRITEO, dev, 1)[0]
for _ in range(3):
    eng.train_step(bx[0], bx[1], bx[2], 1e-3, choice=choice)
torch.cuda.synchronize()
cp = eng.compile(choice, B, train=True)
names = ["loads, parameters parked, barrier", "forward recomputed, q / k / v rows parked", "LayerNorm 2 (+ dW2 partial)", "FFN 2 (+ dW1 partial)", "FFN 1", "LayerNorm 1 (+ dWout partial)",
         "out-projection, dO rows parked, barrier", "attention A (dq)", "attention B (dk, dv)", "token sums, dWin partials, dx", "barrier, partials parked, barrier", "sums over the blocks, stores"]
fnames = ["loads, parameters parked, barrier", "in-projection, k / v rows parked, barrier", "attention: row maxima", "attention: softmax and P V",
          "o and softmax statistics saved", "out-projection, LayerNorm 1", "FFN", "LayerNorm 2", "output + LayerNorm statistics stored"]
mha = []
for d in cp.fwd.descs:
    mha += [n.desc for n in d.nodes] if isinstance(d, L.WorklistDesc) else [d]
for d in mha:
    if d.kind != L.OP_MHA_FWD:
        continue
    raw = torch.as_tensor(_Raw(d.out, B * d.ldo), device=dev).cpu().numpy().astype(np.int64).reshape(B, d.ldo)[::4, :10] & 0xffffffff  # (samples whose wave 0 owns token block 0)
    dt = ((raw[:, 1:] - raw[:, :-1]) & 0xffffffff) / 100.0
    print("MHA_FWD N=%d: whole body median %.2f (x100 shader clocks)" % (d.N, np.median(((raw[:, 9] - raw[:, 0]) & 0xffffffff) / 100.0)))
    for i, n in enumerate(fnames):
        print("   %-46s %6.2f" % (n, np.median(dt[:, i])))
nodes = []
for d in cp.bwd.descs:
    nodes += list(d.nodes) if isinstance(d, L.WorklistDesc) else []
for n in nodes:
    d = n.desc
    if d.kind != L.OP_MHA_BWD:
        continue
    ld = d.partial_ld if d.partial_ld > 0 else L.MHA_PARAMS
    raw = torch.as_tensor(_Raw(d.dparams_partial, B * ld), device=dev).cpu().numpy().astype(np.int64).reshape(B, ld)[::4, :13] & 0xffffffff
    dt = ((raw[:, 1:] - raw[:, :-1]) & 0xffffffff) / 100.0
    print("MHA_BWD N=%d: whole body median %.2f (max %.2f) (x100 shader clocks)" % (d.N, np.median(((raw[:, 12] - raw[:, 0]) & 0xffffffff) / 100.0), np.max(((raw[:, 12] - raw[:, 0]) & 0xffffffff) / 100.0)))
    for i, nm in enumerate(names):
        print("   %-38s %6.2f" % (nm, np.median(dt[:, i])))
    # the same descriptor three times back to back on its own: its planes were read a few microseconds ago (Infinity Cache warm)
    sp = eng.stream.cuda_stream
    b = S.item_bytes(n)
    one = L.WorklistDesc()
    one.kind, one.n = L.OP_WORKLIST, 1
    it = one.item[0]
    it.kind, it.part, it.off = d.kind, S._PART[n.part], 0
    C.memmove(C.addressof(one) + L.WorklistDesc.blob.offset, b, len(b))
    with torch.cuda.stream(eng.stream):
        us = bench.time_desc(lib, L, sp, one, iters=3) * 1e3
    torch.cuda.synchronize()
    raw = torch.as_tensor(_Raw(d.dparams_partial, B * ld), device=dev).cpu().numpy().astype(np.int64).reshape(B, ld)[::4, :13] & 0xffffffff
    dt = ((raw[:, 1:] - raw[:, :-1]) & 0xffffffff) / 100.0
    print("   back to back on its own (%.2f us per launch): whole body %.2f, loads + first barrier %.2f" % (us, np.median(((raw[:, 12] - raw[:, 0]) & 0xffffffff) / 100.0), np.median(dt[:, 0])))
