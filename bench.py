#!/usr/bin/env python3
"""Benchmark of the NASRec hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): Criteo best-1shot sub-network (ea_criteo_kaggle_xlarge_best_1shot.json), batch 256
per GPU, FULL embedding tables (33 762 601 rows, 2.16 GB), fp32.  One "step" = the reference's training step
(train_utils.py:262-286, 386): forward -> BCEWithLogits -> backward -> [gradient exchange] -> clip_grad_norm_(5.0) ->
Adagrad(eps=1e-2) -> LR-schedule update, on a synthetic Criteo-shaped batch that is already resident in HBM.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B = 256
LR_MAX, LR_MIN = 0.16, 1e-8
TRAIN_LIMIT = 36672495  # main_train.py:354
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E spec peak
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak


def synthetic_batches(n, Fd, tables, device, seed):
    """int_x = log(U_int[0,1000)+1), ids uniform per table (cache-hostile), y ~ Bernoulli(0.25)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        int_x = torch.log(torch.randint(0, 1000, (B, Fd), generator=g).float() + 1.0)
        cat_x = torch.stack([torch.randint(0, int(t), (B,), generator=g) for t in tables], dim=1)
        y = (torch.rand(B, generator=g) < 0.25).float()
        out.append((int_x.to(device), cat_x.to(device), y.to(device)))
    return out


def gemm_flops(d):
    f = 0
    for q in range(d.nseg):
        s = d.seg[q]
        if s.A:
            f += 2.0 * s.M * s.N * s.K
    return f


def time_desc(lib, L, stream_ptr, desc, iters=200):
    """average duration of ONE descriptor launch, HIP events on the engine's own stream"""
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.check(lib.nasrec_event_create(C.byref(e0)))
    L.check(lib.nasrec_event_create(C.byref(e1)))
    for _ in range(10):
        L.check(lib.nasrec_launch(stream_ptr, C.addressof(desc)))
    L.check(lib.nasrec_event_record(e0, stream_ptr))
    for _ in range(iters):
        L.check(lib.nasrec_launch(stream_ptr, C.addressof(desc)))
    L.check(lib.nasrec_event_record(e1, stream_ptr))
    ms = C.c_float()
    L.check(lib.nasrec_event_elapsed_ms(e0, e1, C.byref(ms)))
    lib.nasrec_event_destroy(e0)
    lib.nasrec_event_destroy(e1)
    return ms.value / iters


def cpu_baseline(choice, tables, threads):
    """The oracle (CPU restatement of the reference step, dense-gradient semantics exactly like the reference) timed on
    this box's host cores on a bounded sample: a few full B=256 steps with the full tables."""
    from oracle import nasrec_oracle as O
    torch.set_num_threads(threads)
    cfg = O.NetCfg(7, O.ops_config_lib["xlarge"], False, "relu", fixed=True)
    P = O.Params(torch.float32)
    g = torch.Generator().manual_seed(0)
    for f, n in enumerate(tables):  # fast init for the big tables (seeded_param would spend minutes in numpy)
        P["_embedding.%d.weight" % f] = torch.randn(n, 16, generator=g) * (2.0 / (n + 16)) ** 0.5
    int_x, cat_x, y = [t.cpu() for t in synthetic_batches(1, 13, tables, "cpu", 99)[0]]
    y = y.view(-1, 1)
    O.supernet_forward(P, cfg, int_x, cat_x, choice)  # lazy creation of the dense parameters
    P.frozen = True
    state = {}
    O.train_step(P, state, cfg, choice, int_x, cat_x, y, lr=1e-3)  # warm-up (allocates Adagrad state)
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 12.0 and n < 20):
        O.train_step(P, state, cfg, choice, int_x, cat_x, y, lr=1e-3)
        n += 1
        if time.perf_counter() - t0 > 25.0:
            break
    dt = time.perf_counter() - t0
    return dict(value=B * n / dt, unit="samples/s", cores=threads, kind="port",
                sample="%d full training steps (B=256, full 33.76M-row tables, dense-gradient semantics) of the CPU oracle, %.1f s" % (n, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--force-dp-path", action="store_true", help="run the N>1 exchange code path on a single rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_dp_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from nasrec_amd import _lib as L
    from nasrec_amd import plan as P
    from nasrec_amd.engine import SupernetEngine
    from nasrec_amd.parallel import DataParallelStep
    from nasrec_amd.search_space import ops_config_lib
    from nasrec_amd.utils.config import NUM_EMBEDDINGS_CRITEO
    from nasrec_amd.utils.lr_schedule import CosineAnnealingWarmupRestarts
    lib = L.load()

    choice_all = json.load(open(os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
    choice = {"macro": choice_all["macro"], "micro": choice_all["micro"]}
    tables = NUM_EMBEDDINGS_CRITEO
    # main_train.py:258-269: best-1shot sub-networks are built WITHOUT LayerNorm (use_layernorm hard-coded False)
    cfg = P.NetConfig(choice_all["num_blocks"], ops_config_lib[choice_all["config"]], False, "relu", fixed=True)
    eng = SupernetEngine(cfg, 13, 26, tables, device=device, warm_choice=choice, world_size=world)
    eng.init_weights(seed=0)
    batches = synthetic_batches(16, 13, tables, device, 1234 + rank)
    steps_per_epoch = TRAIN_LIMIT // B
    sched = CosineAnnealingWarmupRestarts(steps_per_epoch, max_lr=LR_MAX, min_lr=LR_MIN, warmup_steps=steps_per_epoch // 10)
    dp = DataParallelStep(eng, choice, B, clip=5.0, eps=1e-2, graph=not args.no_graph, force_exchange=args.force_dp_path)

    def run(n, start):
        for i in range(n):
            bx = batches[(start + i) % len(batches)]
            dp.step(bx[0], bx[1], bx[2], sched.get_lr())
            sched.step()

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    run(args.warmup, 0)
    fence()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss = float(dp.cp.loss.item())
    if not (loss == loss):
        raise SystemExit("loss is NaN")

    result = {
        "metric": "supernet samples/sec at batch 256 (Criteo-shape), 1/2/4/8 MI355X",
        "value": B * world * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "Criteo best-1shot sub-network (ea_criteo_kaggle_xlarge_best_1shot.json), full training step, "
                               "batch 256 per GPU, full embedding tables (33.76M rows), fp32",
                   "per_gpu_batch": B, "global_batch": B * world, "graph": dp.graph,
                   "parallelism": "dp%d (RCCL all-reduce of dense grads + all-gather of row-sparse embedding grads)" % world,
                   "embedding_update": "row-sparse clip+Adagrad (== dense reference update for weight_decay 0)"},
        "final_loss": loss,
    }

    if rank == 0 and world == 1:
        # ---- forward-only throughput (eval path) ---------------------------------------------------------------
        fcp = eng.compile(choice, B, train=False, graph=True)
        for _ in range(20):
            eng.forward(batches[0][0], batches[0][1], graph=True)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for i in range(200):
            eng.forward(batches[i % 16][0], batches[i % 16][1], graph=True)
        torch.cuda.synchronize(device)
        result["forward_only_samples_per_s"] = B * 200 / (time.perf_counter() - t1)

        # ---- roofline of the dominant kernel: the largest GEMM launch of the step (fp32 MFMA bound) -----------------
        sp = eng.stream.cuda_stream
        cp = dp.cp
        allg = [d for d in P.iter_ops(cp.fwd.descs + cp.bwd.descs) if isinstance(d, L.GemmDesc)]
        dom = max(allg, key=gemm_flops)
        with torch.cuda.stream(eng.stream):
            ms = time_desc(lib, L, sp, dom)
        fl = gemm_flops(dom)
        s0 = dom.seg[0]
        result["roofline"] = {"bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                              "kernel": "gemm_kernel<%d,%d,%d> M=%d N=%d K=%d nseg=%d splitk=%d" % (
                                  dom.amode, dom.bmode, dom.cmode, s0.M, s0.N, sum(dom.seg[q].K for q in range(dom.nseg)), dom.nseg, dom.splitk),
                              "flops_per_launch": fl, "avg_launch_us": ms * 1e3}
        # HBM/fabric traffic of that launch: PMC counters cannot be collected from inside this process, so the value is the
        # committed rocprofv3 measurement of exactly this launch (profiles/r01_dominant_gemm_traffic.json: FETCH_SIZE x2
        # per the gfx950 correction + WRITE_SIZE, separate --pmc passes); null if the dominant launch is a different one
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_dominant_gemm_traffic.json")))
            if (s0.M, s0.N, sum(dom.seg[q].K for q in range(dom.nseg)), dom.splitk) == (256, 768, 1565, 5):
                result["roofline"]["traffic"] = tj["traffic_bytes_per_launch"]
                result["roofline"]["traffic_unit"] = "bytes per launch (PMC, see profiles/r01_dominant_gemm_traffic.json)"
                result["roofline"]["algorithmic_bytes"] = tj["algorithmic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        # whole-step accounting: executed GEMM FLOPs per step and the HBM bytes the step must move at minimum
        step_flops = sum(gemm_flops(d) for d in allg)
        result["step_gemm_gflop"] = step_flops / 1e9
        result["step_mfma_frac"] = step_flops / (dt / args.steps) / 1e12 / MFMA_F32_PEAK_TFLOPS
        if not args.no_cpu_baseline:
            # torch CPU ops stop scaling (and start thrashing) far below this box's logical core count: use one
            # socket's worth of threads at most, and state the number
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            result["cpu_baseline"] = cpu_baseline(choice, tables, int(os.environ.get("NASREC_CPU_THREADS", min(avail, 32))))
            result["cpu_baseline"]["host_logical_cpus"] = avail
            result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
