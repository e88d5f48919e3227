import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_supernet_fullsize_gpu import _build, _jsonable
from nasrec_amd import _lib as L, plan as P
model, c, ds, tables, int_x, cat_x, y = _build("cfg5_kdd_autoctr_b8192", seed=1)
eng = model._engine
for step in range(3):
    ch = _jsonable(model._resolve_choice(None))
res = {}
for mode in ("useful", "nouseful", "nodefer"):
    os.environ.pop("NASREC_NO_USEFUL", None)
    if mode == "nouseful":
        os.environ["NASREC_NO_USEFUL"] = "1"
    eng._plans.clear()
    cp = eng.compile(ch, c["B"], True, 5.0, 1e-2, graph=False, defer_dw=(mode != "nodefer"))
    sp = eng._sp()
    eng._stage_inputs(sp, cp, int_x, cat_x, y, 0.01)
    cp.fwd.run(sp); cp.bwd.run(sp)
    torch.cuda.synchronize()
    res[mode] = (eng.flat_g.clone(), cp)
    print(mode, "kernels:", sorted({P.gemm_kernel_name(d) + " S=%d" % d.splitk for d in cp.bwd.descs if isinstance(d, L.GemmDesc) and d.zmode and d.amode == 1}))
names = [n for n in res["useful"][1].ctx.grad_params if not n.startswith("_embedding.")]
for a, b in (("useful", "nouseful"), ("useful", "nodefer"), ("nouseful", "nodefer")):
    out = []
    for n in names:
        o, m = eng.offsets[n], eng.params[n].numel()
        ga, gb = res[a][0][o:o + m], res[b][0][o:o + m]
        d = float((ga - gb).abs().max()); sc = float(gb.abs().max())
        if d > 1e-5 * sc:
            out.append((d / max(sc, 1e-30), n, tuple(eng.params[n].shape)))
    print(a, "vs", b, sorted(out, reverse=True)[:12])

# fp64 oracle gradients of the same step (full batch), dense parameters
from oracle import nasrec_oracle as O
torch.set_num_threads(32)
Fs = ds["Fs"]
Pm = O.Params(torch.float64)
cp = res["useful"][1]
for k in cp.used_params:
    if not k.startswith("_embedding."):
        Pm[k] = eng.params[k].detach().double().cpu()
cat_small = torch.zeros(c["B"], Fs, dtype=torch.int64)
for f in range(Fs):
    ids, inv = torch.unique(cat_x[:, f].cpu(), return_inverse=True)
    Pm["_embedding.%d.weight" % f] = eng.tables[f][ids.cuda()].double().cpu()
    cat_small[:, f] = inv
leaves = {k: v.detach().requires_grad_(True) for k, v in Pm.items()}
Pl = O.Params(torch.float64, frozen=True)
Pl.update(leaves)
ocfg = O.NetCfg(7, O.ops_config_lib[c["space"]], True, "relu", fixed=False)
import time
t0 = time.time()
logits = O.supernet_forward(Pl, ocfg, int_x.double().cpu(), cat_small, ch)
loss = O.bce_with_logits_mean(logits.view(-1), y.double().cpu())
ks = [k for k in leaves if not k.startswith("_embedding.")]
gr = torch.autograd.grad(loss, [leaves[k] for k in ks], allow_unused=True)
print("oracle step %.1f s" % (time.time() - t0))
for mode in res:
    out = []
    for k, g in zip(ks, gr):
        if g is None:
            continue
        o, m = eng.offsets[k], eng.params[k].numel()
        ge = res[mode][0][o:o + m].double().cpu()
        d = float((ge - g.reshape(-1)).abs().max()); sc = float(g.abs().max())
        out.append((d / max(sc, 1e-30), k))
    print(mode, "vs oracle worst:", sorted(out, reverse=True)[:5])
