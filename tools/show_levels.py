#!/usr/bin/env python3
"""Level structure (nasrec_amd/schedule.py) of the batch-256 training step of a fixed sub-network, built on the CPU (descriptors
only, nothing is launched): which operators run side by side in each heterogeneous launch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nasrec_amd import _lib as L, plan as P, schedule as S
from nasrec_amd.search_space import ops_config_lib

def build_cpu_plan(cfg_json, B, Fd, Fs, defer_dw=False):
    ca = json.load(open(cfg_json))
    choice = {"macro": ca["macro"], "micro": ca["micro"]}
    cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
    shapes = P.infer_param_shapes(cfg, choice, Fd, Fs, [10] * Fs)
    params = {k: torch.zeros(v) for k, v in shapes.items() if not k.startswith("_embedding.")}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    ctx = P.Ctx(B, "cpu", params, grads, train=True)
    ctx.defer_dw = defer_dw
    int_buf = P.Buf(ctx, B * Fd, need_grad=False)
    sbuf = ctx.buf(B * Fs * 16)
    ctx.raw_sparse = sbuf
    (d_last,), (s_last,) = P.network_walk(ctx, cfg, choice, P.DV(int_buf, 0, Fd, Fd), P.SV(sbuf, 0, Fs, Fs * 16))
    K = d_last.width + s_last.N * 16
    w = ctx.param("_final.weight", (1, K)); bptr = ctx.param("_final.bias", (1,))
    logits, dlog = ctx.alloc(B), ctx.alloc(B)
    fsegs = [P.Seg(d_last, 0, d_last.width), P.Seg(s_last.dense(), d_last.width, s_last.N * 16)]
    fd = L.FinalDesc(); fd.kind = L.OP_FINAL_FWD; fd.B, fd.nseg = B, 2
    fd.w, fd.bias, fd.logits = w, bptr, logits.data_ptr()
    for q, s in enumerate(fsegs):
        fd.seg[q], fd.width[q], fd.ld[q], fd.off[q] = s.view.ptr, s.width, s.view.ld, s.koff
    ctx.emit(fd)
    def final_bwd():
        e = L.FinalDesc(); e.kind = L.OP_FINAL_BWD; e.B, e.nseg = B, 2
        e.w, e.bias, e.dlogits = w, bptr, dlog.data_ptr()
        e.dw, e.dbias = grads["_final.weight"].data_ptr(), grads["_final.bias"].data_ptr()
        for q, s in enumerate(fsegs):
            gp, acc = ctx.gtarget(s.view)
            e.seg[q], e.dseg[q], e.width[q], e.ld[q], e.off[q], e.dseg_accumulate[q] = s.view.ptr, gp, s.width, s.view.ld, s.koff, acc
        ctx.emit(e)
    ctx.on_backward(final_bwd)
    ctx.build_backward()
    return ctx

def describe(n):
    d = n.desc
    names = {getattr(L, k): k[3:] for k in dir(L) if k.startswith("OP_")}
    if isinstance(d, L.GemmDesc):
        segs = [(d.seg[q].M, d.seg[q].N, d.seg[q].K) for q in range(d.nseg) if d.seg[q].A]
        return "GEMM[%s] a%d b%d c%d z%d S%d %s" % (n.part, d.amode, d.bmode, d.cmode, d.zmode, d.splitk, segs[:3] + (["..."] if len(segs) > 3 else []))
    return names.get(d.kind, str(d.kind))

if __name__ == "__main__":
    cfgp = os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")
    ctx = build_cpu_plan(cfgp, 256, 13, 26, defer_dw=("--defer" in sys.argv))
    for name, prog in (("forward", ctx.fwd), ("backward", ctx.bwd)):
        lv = S.levels_of(prog)
        print("%s: %d descriptors -> %d nodes in %d levels" % (name, len(prog), sum(len(x) for x in lv), len(lv)))
        for i, nodes in enumerate(lv):
            print("  L%-2d %s" % (i, " || ".join(describe(n) for n in nodes)))
