import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from helpers import load_golden
from test_parity_gpu import build_engine
from nasrec_amd import _lib as L, plan as P
name = sys.argv[1] if len(sys.argv) > 1 else "fixed_kdd_autoctr"
z, meta = load_golden(os.path.join("tests", "golden", name + ".npz"))
res = {}
for mode in ("defer", "plain"):
    os.environ.pop("NASREC_NO_RAW_DEFER", None)
    if mode == "plain":
        os.environ["NASREC_NO_RAW_DEFER"] = "1"
    eng = build_engine(z, meta)
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda()
    cp = eng.forward_backward(int_x, cat_x, y, meta["choice"])
    torch.cuda.synchronize()
    g = cp.sparse0.grad_tensor().view(int_x.shape[0], -1, 16).clone()
    res[mode] = g
    descs = [d for d in cp.bwd.descs if isinstance(d, L.GemmDesc) and d.amode == L.AM_RC and d.bmode == L.AM_TOKR]
    print(mode, "raw-ish dx launches:", [(d.zmode, d.nseg, d.beta, [(d.seg[q].M, d.seg[q].K, d.seg[q].lda, d.seg[q].ldb, d.seg[q].ldc, bool(d.seg[q].Baux), d.seg[q].accumulate) for q in range(d.nseg)]) for d in descs])
d = (res["defer"] - res["plain"]).abs()
print("max diff", float(d.max()), "per token", d.amax((0, 2)).tolist())
print("per sample (first 8)", d.amax((1, 2))[:8].tolist())
