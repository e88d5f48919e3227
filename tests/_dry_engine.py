"""Stand-in engine for the DRY RUN of `bench.py --gpus N` on a box without GPUs (tests/test_bench_dry_gloo_cpu.py): the data-parallel
protocol of nasrec_amd/parallel.py on CPU tensors with the fp64 oracle inside — the OracleEngine of tests/test_data_parallel_cpu.py, fed
by bench.py's fp32 batches.  Test infrastructure only (it imports oracle/): bench.py loads it through NASREC_BENCH_DRY_ENGINE and never
on a measured run."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)


def make(choice_all, choice, Fd, Fs, tables):
    from oracle import nasrec_oracle as O
    from test_data_parallel_cpu import OracleEngine

    class DryEngine(OracleEngine):
        Fd_ = Fd

        def dp_plan(self, *a, **kw):
            plan = OracleEngine.dp_plan(self, *a, **kw)
            stage = plan.stage

            def stage64(int_x, cat_x, y, lr):
                self.lr = float(lr)
                stage(int_x.double(), cat_x, y.double(), lr)
            plan.stage = stage64
            return plan

    cfg = O.NetCfg(choice_all["num_blocks"], O.ops_config_lib[choice_all["config"]], False, "relu", fixed=True)
    P = O.Params(torch.float64)
    int_x, cat_x, _ = O.synthetic_batch(4, Fd, tables, seed=3)
    with torch.no_grad():
        O.supernet_forward(P, cfg, int_x.double(), cat_x, choice, num_embeddings=tables)
    P.frozen = True
    eng = DryEngine(cfg, P, Fs)
    eng.Fd = Fd
    eng.lr = 1e-3
    return eng
