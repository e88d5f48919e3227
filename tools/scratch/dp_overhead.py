"""Where the N > 1 code path loses time against the plain step on one GPU (cfg 2): variants of the exchange, world = 1."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.distributed as dist
import bench
from nasrec_amd import plan as P, parallel as PAR
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.utils.config import DATASETS
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29561", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
ds = DATASETS["criteo"]; tables = ds["tables"]; B = 256
ca = json.load(open("nasrec_amd/configs/criteo/ea_criteo_kaggle_xlarge_best_1shot.json"))
choice = {"macro": ca["macro"], "micro": ca["micro"]}
cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
batches = bench.synthetic_batches(16, B, ds["Fd"], tables, dev, 1234)
def run(tag, force, mod=None):
    eng = SupernetEngine(cfg, ds["Fd"], ds["Fs"], tables, device=dev, warm_choice=choice)
    eng.init_weights(seed=0)
    dp = PAR.DataParallelStep(eng, choice, B, clip=5.0, eps=1e-2, graph=True, force_exchange=force)
    if mod: mod(dp)
    for i in range(30): dp.step(*batches[i % 16], 1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300): dp.step(*batches[i % 16], 1e-3)
    torch.cuda.synchronize(); print("%-44s %.1f us/step" % (tag, (time.perf_counter() - t0) / 300 * 1e6))
    del eng, dp; torch.cuda.empty_cache()
run("plain", False)
run("exchange (as shipped)", True)

def make_step(skip_ids=False, skip_rows=False, skip_ar=False):
    def step(self, int_x, cat_x, y, lr, choice=None):
        eng = self.engine
        plan = self._plan(self.choice)
        plan.stage(int_x, cat_x, y, lr)
        pending = [] if skip_ids else [PAR.all_gather_rows_async(self.cat_all, plan.cat_local)]
        plan.forward()
        for runseg, ranges in plan.segments:
            runseg()
            if not skip_ar:
                for off, n in ranges:
                    pending.append(dist.all_reduce(eng.flat_g[off:off + n], op=dist.ReduceOp.SUM, async_op=True))
        if not skip_rows:
            pending.append(PAR.all_gather_rows_async(self.sg_all, plan.sparse_grad))
        for w in pending:
            if w is not None:
                w.wait()
        self.opt(plan)
        self._last = ("dp", plan)
        return plan.loss
    return step

import types
for tag, kw in (("no ids all-gather", dict(skip_ids=True)), ("no row-grad all-gather", dict(skip_rows=True)), ("no all-reduce", dict(skip_ar=True)),
                ("no collectives at all (3 graphs)", dict(skip_ids=True, skip_rows=True, skip_ar=True))):
    run(tag, True, lambda dp, kw=kw: setattr(dp, "step", types.MethodType(make_step(**kw), dp)))

def step_at_end(self, int_x, cat_x, y, lr, choice=None):
    eng = self.engine
    plan = self._plan(self.choice)
    plan.stage(int_x, cat_x, y, lr)
    plan.forward()
    for runseg, ranges in plan.segments:
        runseg()
    pending = [PAR.all_gather_rows_async(self.cat_all, plan.cat_local), PAR.all_gather_rows_async(self.sg_all, plan.sparse_grad),
               dist.all_reduce(eng.flat_g, op=dist.ReduceOp.SUM, async_op=True)]
    for w in pending:
        if w is not None:
            w.wait()
    self.opt(plan)
    self._last = ("dp", plan)
    return plan.loss

def step_sync_ops(self, int_x, cat_x, y, lr, choice=None):
    eng = self.engine
    plan = self._plan(self.choice)
    plan.stage(int_x, cat_x, y, lr)
    plan.forward()
    for runseg, ranges in plan.segments:
        runseg()
    PAR.all_gather_rows(self.cat_all, plan.cat_local)
    PAR.all_gather_rows(self.sg_all, plan.sparse_grad)
    dist.all_reduce(eng.flat_g, op=dist.ReduceOp.SUM)
    self.opt(plan)
    self._last = ("dp", plan)
    return plan.loss

run("all three collectives at the end (async)", True, lambda dp: setattr(dp, "step", types.MethodType(step_at_end, dp)))
run("all three at the end (blocking API)", True, lambda dp: setattr(dp, "step", types.MethodType(step_sync_ops, dp)))
