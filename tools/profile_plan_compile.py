#!/usr/bin/env python3
"""Host cost of compiling one sampled supernet path (plan.py walk + descriptor fill), measured on the CPU: descriptors only, nothing is
launched.  `python tools/profile_plan_compile.py [xlarge|autoctr] [--prof]`"""
import cProfile, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nasrec_amd import _lib as L, plan as P
from nasrec_amd.engine import Arena
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.supernet.supernet import SuperNet

space = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "xlarge"
B, Fd, Fs = 4096, 13, 26
cfg = P.NetConfig(7, ops_config_lib[space], True, "relu", fixed=False)
full = P.full_path_choice(cfg)
shapes = P.infer_param_shapes(cfg, full, Fd, Fs, [10] * Fs)
off, offsets = 0, {}
for n, shp in shapes.items():
    if n.startswith("_embedding."):
        continue
    offsets[n] = off
    off += (int(np.prod(shp)) + 3) // 4 * 4
flat_p = torch.empty(off)
flat_g = torch.empty(off)
params = {n: flat_p[o:o + int(np.prod(shapes[n]))].view(shapes[n]) for n, o in offsets.items()}
grads = {n: flat_g[o:o + int(np.prod(shapes[n]))].view(shapes[n]) for n, o in offsets.items()}
arena = Arena("cpu")
PCACHE = {}
# paths from the drop-in module's own sampler (host-side bookkeeping: no engine, no GPU)
_m = SuperNet(num_blocks=7, ops_config=ops_config_lib[space], use_layernorm=True, num_embeddings=[10] * Fs, sparse_input_size=Fs,
              path_sampling_strategy="default", fixed=False, anypath_choice="binomial-0.5")
np.random.seed(0)


class sampler:
    sample = staticmethod(lambda: _m._resolve_choice(None))


def compile_once(choice):
    arena.reset()
    ctx = P.Ctx(B, "cpu", params, grads, train=True)
    ctx.arena = arena
    ctx.defer_dw = True
    int_buf = P.Buf(ctx, B * Fd, need_grad=False)
    sbuf = ctx.buf(B * Fs * 16)
    ctx.raw_sparse = sbuf
    ctx._pcache = PCACHE
    (d_last,), (s_last,) = P.network_walk(ctx, cfg, choice, P.DV(int_buf, 0, Fd, Fd), P.SV(sbuf, 0, Fs, Fs * 16))
    fsegs = [P.Seg(d_last, 0, d_last.width), P.Seg(s_last.dense(), d_last.width, s_last.N * 16)]

    def final_bwd():  # stands in for the engine's final-logit backward: makes the last block's outputs live
        e = L.FinalDesc()
        e.kind = L.OP_FINAL_BWD
        for q, sgm in enumerate(fsegs):
            gp, acc = ctx.gtarget(sgm.view)
            e.seg[q], e.dseg[q] = sgm.view.ptr, gp
        ctx.emit(e)

    ctx.on_backward(final_bwd)
    ctx.build_backward()
    return len(ctx.fwd), len(ctx.bwd)


choices = [json.loads(json.dumps(sampler.sample(), default=lambda o: o.tolist() if hasattr(o, "tolist") else o.item())) for _ in range(40)]
compile_once(choices[0])
ts = []
for c in choices:
    t0 = time.perf_counter()
    n = compile_once(c)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("%s: compile of one sampled path: mean %.2f ms, median %.2f, max %.2f (%d + %d launches in the last one)" % (space, ts.mean(), np.median(ts), ts.max(), n[0], n[1]))
if "--prof" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for c in choices[:20]:
        compile_once(c)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
