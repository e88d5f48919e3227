import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.distributed as dist
from tests.test_supernet_fullsize_gpu import _build, _jsonable
from nasrec_amd.parallel import DataParallelStep
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29548", rank=0, world_size=1, device_id=torch.device("cuda", 0))
runs = []
for force in (False, True):
    model, c, ds, tables, int_x, cat_x, y = _build("cfg5_kdd_autoctr_b8192", seed=1)
    eng = model._engine
    dp = DataParallelStep(eng, None, c["B"], clip=5.0, eps=1e-2, graph=False, force_exchange=force)
    snaps = []
    for _ in range(3):
        ch = _jsonable(model._resolve_choice(None))
        dp.step(int_x, cat_x, y, 0.01, choice=ch)
        torch.cuda.synchronize()
        snaps.append((eng.flat_p.clone(), eng.flat_g.clone(), eng.clip_out.clone()))
    runs.append((snaps, dict(eng.offsets), {n: eng.params[n].numel() for n in eng.dense_names}))
    del model, eng, dp
    torch.cuda.empty_cache()
(a, offs, nums), (b, _, _) = runs
for s in range(3):
    dpar = (a[s][0] - b[s][0]).abs()
    i = int(dpar.argmax())
    name = next(n for n in offs if offs[n] <= i < offs[n] + nums[n])
    print("step", s, "max |dp|", float(dpar.max()), "at", name, i - offs[name], "p", float(a[s][0][i]), float(b[s][0][i]),
          "g", float(a[s][1][i]), float(b[s][1][i]), "clip", a[s][2].tolist(), b[s][2].tolist())
    # per-parameter worst offenders
    worst = []
    for n in offs:
        o, m = offs[n], nums[n]
        d = float(dpar[o:o + m].max())
        if d > 0:
            worst.append((d, n))
    print("   ", sorted(worst, reverse=True)[:6])
dist.destroy_process_group()
