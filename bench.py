#!/usr/bin/env python3
"""Benchmark of the NASRec hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py [--config {2,3,4,5}] [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = the reference's training step (train_utils.py:262-286, 386): forward -> BCEWithLogits -> backward ->
[gradient exchange] -> clip_grad_norm_(5.0) -> Adagrad(eps=1e-2) -> LR-schedule update, on a synthetic batch that is already
resident in HBM.  Workloads = BASELINE.json `configs`:
  2 (default, the config the headline metric is quoted on)  Criteo best-1shot sub-network, batch 256 per GPU, full tables;
  3  Criteo NASRec-Full (xlarge) supernet, LayerNorm, `default` path sampling / binomial-0.5, batch 4096, full tables;
  4  Avazu NASRec-Full supernet, batch 4096 per GPU;      5  KDD NASRec-Small (autoctr) supernet, 0.5M-capped tables, batch 8192 per GPU.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LR_MAX, LR_MIN = 0.16, 1e-8
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak

WORKLOADS = {
    2: dict(dataset="criteo", mode="fixed", B=256, train_limit=36672495,  # main_train.py:354
            name="Criteo best-1shot sub-network (ea_criteo_kaggle_xlarge_best_1shot.json), full training step, batch 256 per GPU, "
                 "full embedding tables (33.76M rows), fp32"),
    3: dict(dataset="criteo", mode="supernet", space="xlarge", B=4096, cap=None, train_limit=36672495,
            name="Criteo NASRec-Full (xlarge) supernet, 7 blocks, LayerNorm, path sampling `default` / binomial-0.5 (one path per step "
                 "from np.random), full training step, batch 4096 per GPU, full embedding tables (33.76M rows), fp32"),
    4: dict(dataset="avazu", mode="supernet", space="xlarge", B=4096, cap=None, train_limit=32343175,
            name="Avazu NASRec-Full (xlarge) supernet, 7 blocks, LayerNorm, path sampling `default` / binomial-0.5, full training step, "
                 "batch 4096 per GPU, full embedding tables (9.46M rows), fp32"),
    5: dict(dataset="kdd", mode="supernet", space="autoctr", B=8192, cap=500000, train_limit=119711284,
            name="KDD-Cup'12 NASRec-Small (autoctr) supernet, 7 blocks, LayerNorm, path sampling `default` / binomial-0.5, full training "
                 "step, batch 8192 per GPU, tables capped at 0.5M rows (3.04M rows), fp32"),
}


def synthetic_ids(B, tables, g, dist_name="uniform"):
    """ids of one batch, [B, Fs] int64.  uniform: U_int[0, n_f) per table — the cache-hostile primary case of SURVEY §8d;
    zipf: Zipf(alpha = 1.05) over the ranks 1 .. n_f - 1 (bounded, by inverse CDF) with 2 % zeros (0 = missing id,
    data_pipes.py:141,164); rank r lands on id 1 + (r * 2654435761) mod (n_f - 1), as the reference's hash of the raw token does."""
    if dist_name == "uniform":
        return torch.stack([torch.randint(0, int(t), (B,), generator=g) for t in tables], dim=1)
    cols = []
    a = 1.05
    for t in tables:
        n = max(int(t) - 1, 1)
        u = torch.rand(B, generator=g, dtype=torch.float64)
        rank = torch.floor(((float(n) ** (1.0 - a) - 1.0) * u + 1.0) ** (1.0 / (1.0 - a))).clamp_(1, n).to(torch.int64)
        ids = 1 + (rank * 2654435761) % n
        ids[torch.rand(B, generator=g) < 0.02] = 0
        cols.append(ids.clamp_(0, int(t) - 1))
    return torch.stack(cols, dim=1)


def synthetic_batches(n, B, Fd, tables, device, seed, zero_dense=False, ids="uniform"):
    """int_x = log(U_int[0,1000)+1) (Avazu: zeros, data_pipes.py:181), ids per `synthetic_ids`, y ~ Bernoulli(0.25)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        int_x = torch.zeros(B, Fd) if zero_dense else torch.log(torch.randint(0, 1000, (B, Fd), generator=g).float() + 1.0)
        cat_x = synthetic_ids(B, tables, g, ids)
        y = (torch.rand(B, generator=g) < 0.25).float()
        out.append((int_x.to(device), cat_x.to(device), y.to(device)))
    return out


class BatchPool:
    """Pre-generated synthetic batches resident in HBM.  The ids come from a pool large enough that the set of table rows touched
    between two uses of the same id batch exceeds the 256 MB Infinity Cache (cfg 2: 1024 id batches x 256 x 26 rows x 64 B = 436 MB
    of rows, 54 MB of ids; the supernets: 64 batches x 4096 x Fs x 64 B >= 250 MB), so that the gather / dedup / row-Adagrad kernels
    see HBM latency on every step; dense features and labels (a few KB) cycle through a small pool."""

    def __init__(self, n_ids, n_dense, B, Fd, tables, device, seed, zero_dense, ids):
        g = torch.Generator().manual_seed(seed)
        self.dense = synthetic_batches(n_dense, B, Fd, tables[:1], "cpu", seed + 7, zero_dense)
        self.dense = [(a.to(device), c.to(device)) for a, _, c in self.dense]
        self.ids = torch.stack([synthetic_ids(B, tables, g, ids) for _ in range(n_ids)]).to(device)  # [n_ids, B, Fs]
        self.n = n_ids

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        d = self.dense[i % len(self.dense)]
        return d[0], self.ids[i % self.n], d[1]


def gemm_algorithmic_bytes(d):
    """bytes a GEMM launch has to move if every operand element is read once and every output element written once"""
    s0 = d.seg[0]
    if d.zmode:
        return sum(4 * (d.seg[q].M * d.seg[q].K + d.seg[q].N * d.seg[q].K + d.seg[q].M * d.seg[q].N) for q in range(d.nseg) if d.seg[q].A)
    b = sum(4 * (s0.M + s0.N) * d.seg[q].K for q in range(d.nseg) if d.seg[q].A)
    outs = 1 + (1 if d.save_z else 0) + (1 if d.save_act else 0) + (1 if d.beta else 0) + (1 if d.pre_add else 0)
    b += 4 * s0.M * s0.N * outs + (4 * s0.N if d.bias else 0)
    if 1 < d.splitk < 64:
        b += 2 * 4 * d.splitk * s0.M * s0.N
    return b


def gemm_flops(d):
    f = 0
    for q in range(d.nseg):
        s = d.seg[q]
        if s.A:
            f += 2.0 * s.M * (s.N - (1 if s.ones_col else 0)) * s.K
    return f


def time_desc(lib, L, stream_ptr, desc, iters=200):
    """average duration of ONE descriptor launch, HIP events on the stream the kernel is launched on"""
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.check(lib.nasrec_event_create(C.byref(e0)))
    L.check(lib.nasrec_event_create(C.byref(e1)))
    for _ in range(10):
        L.check(lib.nasrec_launch(stream_ptr, C.addressof(getattr(desc, "launch_as", desc))))  # (the form the step launches: a resident worklist descriptor if the plan has one)
    L.check(lib.nasrec_event_record(e0, stream_ptr))
    for _ in range(iters):
        L.check(lib.nasrec_launch(stream_ptr, C.addressof(getattr(desc, "launch_as", desc))))  # (the form the step launches: a resident worklist descriptor if the plan has one)
    L.check(lib.nasrec_event_record(e1, stream_ptr))
    ms = C.c_float()
    L.check(lib.nasrec_event_elapsed_ms(e0, e1, C.byref(ms)))
    lib.nasrec_event_destroy(e0)
    lib.nasrec_event_destroy(e1)
    return ms.value / iters


def launch_kernel_name(L, P, d):
    """the device kernel a descriptor launches (as rocprofv3 names it, template arguments shortened)"""
    if isinstance(d, L.PersistDesc):
        return "persist_kernel<%s>" % ("true" if d.big else "false")
    if isinstance(d, L.WorklistDesc):
        # (csrc/worklist.hip wl_finalize: `big`; the descriptor-resident form of a fixed plan — engine._resident_worklists — is a kernel of its own name)
        return "%s<%s>" % ("worklist_dev_kernel" if hasattr(d, "launch_as") else "worklist_kernel", "true" if any(n.desc.kind == L.OP_MHA_BWD for n in d.nodes) else "false")
    if isinstance(d, L.GemmDesc):
        return P.gemm_kernel_name(d)
    names = {getattr(L, n): n[3:].lower() for n in dir(L) if n.startswith("OP_")}
    return names.get(d.kind, "kind%d" % d.kind) + "_kernel"


def dense_floats_walked(eng, cp):
    """floats of the dense arena the optimizer's launches (square sum, Adagrad) actually walk: the plan's chunk table when it has one
    (the ranges a gradient reaches: engine.compile), else the whole arena"""
    tab = getattr(cp, "chunk_tab", None)
    if tab is None:
        return int(eng.flat_numel)
    return int(tab.view(-1)[1:2 * cp.nchunks:2].sum().item())  # (offset, count) pairs


def launch_work(L, S, d, n_dense=0):
    """(MFMA flops, algorithmic bytes, item summary) of one launch: flops = the products' 2·M·N·K (split-K second passes and the
    vector-ALU bodies — Transformer, FM, DotProduct cores — count 0: the MFMA roof is about the matrix pipe); bytes = every operand
    / result window of every item once (schedule.desc_io, the footprints the level scheduler itself trusts; split-K slabs written by the
    main pass and read back by the second pass included)"""
    nodes = d.nodes if isinstance(d, (L.WorklistDesc, L.PersistDesc)) else [S.Node(d)]
    names = {getattr(L, n): n[3:] for n in dir(L) if n.startswith("OP_")}
    fl, by, items = 0.0, 0, []
    # kinds outside the scheduler's model (they are never reordered): staging + gather, the optimizer's launches
    k = d.kind
    if k == L.OP_STAGE_INPUTS:   # batch copy (read + write) + B x Fs table rows of 64 B read and written
        by = d.B * (d.Fd * 8 + d.Fs * 16 + 8) + d.B * d.Fs * 128
    elif k in (L.OP_OPT_REDUCE, L.OP_EMB_DEDUP):  # ids + row gradients read, summed rows + leaders written (+ the gradient arena read once)
        dd = d.dedup if k == L.OP_OPT_REDUCE else d
        by = dd.B * dd.Fs * (8 + 64 + 64 + 4) + (n_dense * 4 if k == L.OP_OPT_REDUCE else 0)
    elif k == L.OP_OPT_REDUCE2:  # row gradients read once (duplicates' sums written back in place: counted once more), leaders + lists read, the arena's gradients read once
        by = d.B * d.Fs * (64 + 64 + 12) + n_dense * 4
    elif k == L.OP_SUMSQ:
        by = n_dense * 4
    elif k == L.OP_OPT_APPLY:    # dense (n_dense = the floats of the chunk table the launch walks): g read, state and parameter read + written; rows: leader flag + id + summed gradient read, table row + state row r/w
        by = n_dense * 20 + d.rows.B * d.rows.Fs * (12 + 64 + 4 * 64)
    for n in nodes:
        if isinstance(n.desc, L.GemmDesc) and n.part != "epi":
            fl += gemm_flops(n.desc)
        if n.reads is not None:
            by += sum(a.rows * a.width for a in n.reads) + sum(a.rows * a.width for a in n.writes)
        items.append(names.get(n.desc.kind, "?") + ("" if n.part == "whole" else ":" + n.part))
    return fl, by, items


def launch_table(lib, L, P, S, sp, descs, iters, n_dense=0):
    """one row per launch of the step: kernel, items, flops, algorithmic bytes, isolated duration (HIP events on the launch stream,
    back-to-back launches of the same descriptor: cold L2 at every kernel boundary, warm Infinity Cache — what it sees inside the step)"""
    rows = []
    for d in descs:
        fl, by, items = launch_work(L, S, d, n_dense)
        us = time_desc(lib, L, sp, d, iters=iters) * 1e3
        rows.append({"kernel": launch_kernel_name(L, P, d), "items": items, "mflop": fl / 1e6, "alg_KB": by / 1e3, "us": us})
    return rows


def launch_table_in_step(lib, L, sp, descs, reps, blocker, warm=None):
    """duration of every launch of the step IN THE STEP'S OWN ORDER (HIP events on the launch stream between consecutive launches of one
    real step): each kernel sees the caches as its predecessors left them, not the warm state of a back-to-back relaunch of the same
    descriptor.  Ahead of each timed step a blocker (a few ms of unrelated GPU work) lets the host enqueue the whole step before the first
    launch starts, and `warm` (one untimed step) puts the step's own working set back into the Infinity Cache.
    An event between two launches is a barrier packet of its own (~3.5 us on this stack): the SAME sequence is therefore timed once more
    without the inner events (two events around it), and every interval is reduced by the mean excess (sum of intervals - plain sequence)
    / launches — the corrected durations add up to the time the launches take back to back, each including its own launch gap.
    -> (corrected durations in us, raw event intervals in us, us of the plain sequence)"""
    n = len(descs)
    evs = []
    for _ in range(n + 1):
        e = C.c_void_p()
        L.check(lib.nasrec_event_create(C.byref(e)))
        evs.append(e)
    acc = np.zeros(n)
    seq = 0.0
    ms = C.c_float()
    for _ in range(reps):
        blocker()
        if warm is not None:
            warm()
        L.check(lib.nasrec_event_record(evs[0], sp))
        for i, d in enumerate(descs):
            L.check(lib.nasrec_launch(sp, C.addressof(getattr(d, "launch_as", d))))
            L.check(lib.nasrec_event_record(evs[i + 1], sp))
        for i in range(n):
            L.check(lib.nasrec_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms)))
            acc[i] += ms.value
        blocker()
        if warm is not None:
            warm()
        L.check(lib.nasrec_event_record(evs[0], sp))
        for d in descs:
            L.check(lib.nasrec_launch(sp, C.addressof(getattr(d, "launch_as", d))))
        L.check(lib.nasrec_event_record(evs[1], sp))
        L.check(lib.nasrec_event_elapsed_ms(evs[0], evs[1], C.byref(ms)))
        seq += ms.value
    for e in evs:
        lib.nasrec_event_destroy(e)
    raw = acc / reps * 1e3
    seq_us = seq / reps * 1e3
    excess = max(0.0, (float(raw.sum()) - seq_us) / n)
    return np.maximum(raw - excess, 0.25 * raw), raw, seq_us


def csrc_build_id():
    """hash of the kernel sources: a measured artefact under profiles/ is only quoted for the build it was taken from"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "nasrec_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(w, choice_or_sampler, tables, threads, Fd):
    """The oracle (CPU restatement of the reference step, dense-gradient semantics exactly like the reference) timed on this box's
    host cores on a bounded sample of the same workload: a few full steps at the workload's batch size and table sizes."""
    from oracle import nasrec_oracle as O
    torch.set_num_threads(threads)
    B = w["B"]
    fixed = w["mode"] == "fixed"
    if fixed:
        cfg = O.NetCfg(7, O.ops_config_lib["xlarge"], False, "relu", fixed=True)
    else:
        cfg = O.NetCfg(7, O.ops_config_lib[w["space"]], True, "relu", fixed=False)
    P = O.Params(torch.float32)
    g = torch.Generator().manual_seed(0)
    for f, n in enumerate(tables):  # fast init for the big tables (seeded_param would spend minutes in numpy)
        P["_embedding.%d.weight" % f] = torch.randn(n, 16, generator=g) * (2.0 / (n + 16)) ** 0.5
    int_x, cat_x, y = [t.cpu() for t in synthetic_batches(1, B, Fd, tables, "cpu", 99, zero_dense=(w["dataset"] == "avazu"))[0]]
    y = y.view(-1, 1)
    warm = choice_or_sampler if fixed else O.full_path_choice(cfg)
    with torch.no_grad():
        O.supernet_forward(P, cfg, int_x[:64], cat_x[:64], warm, num_embeddings=tables)  # lazy creation of the dense parameters (full path)
    P.frozen = True
    sampler = None
    if not fixed:
        np.random.seed(0)
        sampler = O.PathSampler(cfg, "default", "binomial-0.5")
    state = {}
    pick = (lambda: choice_or_sampler) if fixed else (lambda: sampler.sample())
    O.train_step(P, state, cfg, pick(), int_x, cat_x, y, lr=1e-3)  # warm-up (allocates Adagrad state)
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 12.0 and n < 20):
        O.train_step(P, state, cfg, pick(), int_x, cat_x, y, lr=1e-3)
        n += 1
        if time.perf_counter() - t0 > 25.0:
            break
    dt = time.perf_counter() - t0
    return dict(value=B * n / dt, unit="samples/s", cores=threads, kind="port",
                sample="%d full training steps (B=%d, %d-row tables, dense-gradient semantics%s) of the CPU oracle, %.1f s" % (
                    n, B, sum(tables), "" if fixed else ", one sampled path per step", dt))


def unchanged_harness_route(device, B, tables, Fd, Fs, choice_all, batches, steps=30, warmup=5):
    """The step exactly as the reference's harness writes it (nasrec/utils/train_utils.py:262-286), NOTHING fused:
        optimizer.zero_grad(); out = model(int_x, cat_x); loss = BCEWithLogitsLoss()(out, y); loss.backward();
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); optimizer.step()        # torch.optim.Adagrad(eps=1e-2)
    on the drop-in module with nn.Embedding(sparse=False) semantics (supernet.py:404-410): DENSE .grad tensors on the full tables.  The
    engine runs forward and backward; the table gradients (zero-fill + scatter of B x Fs rows), the norm, the clip multiply and Adagrad
    over all rows are torch's kernels.  Bytes model of SURVEY 8d: per table element 4 B x (grad zero + grad norm read + clip read/write +
    Adagrad: grad read, state read/write, parameter read/write) = 9 passes over the tables."""
    from nasrec_amd.supernet.supernet import SuperNet, ops_config_lib
    torch.manual_seed(0)
    m = SuperNet(num_blocks=choice_all["num_blocks"], ops_config=ops_config_lib[choice_all["config"]], use_layernorm=False, num_embeddings=tables,
                 sparse_input_size=Fs, path_sampling_strategy="fixed-path", fixed=True, fixed_choice=choice_all).to(device)
    with torch.no_grad():
        m(batches[0][0], batches[0][1])
    opt = torch.optim.Adagrad(m.parameters(), lr=1e-3, eps=1e-2)
    loss_fn = torch.nn.BCEWithLogitsLoss()

    def step(i):
        int_x, cat_x, y = batches[i % len(batches)]
        opt.zero_grad()
        loss = loss_fn(m(int_x, cat_x), y.view(-1, 1))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        opt.step()
        return loss
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    n_tab = sum(int(t) for t in tables) * 16
    n_dense = sum(p.numel() for n, p in m.named_parameters() if not n.startswith("_embedding."))
    by = (n_tab + n_dense) * 4 * 9
    del opt, m
    torch.cuda.empty_cache()
    return {"samples_per_s": B / dt, "ms_per_step": dt * 1e3, "final_loss": float(loss),
            "bytes_model_GB": by / 1e9, "achieved_GBps": by / dt / 1e9, "frac_of_hbm_peak": by / dt / 1e9 / HBM_PEAK_GBS,
            "roofline_ms_at_hbm_peak": by / (HBM_PEAK_GBS * 1e9) * 1e3,
            "note": "reference harness step unchanged (train_utils.py:262-286): zero_grad, module forward, BCEWithLogitsLoss, backward with DENSE table "
                    "gradients, torch clip_grad_norm_, torch.optim.Adagrad over every row of every table; bytes model = (table + dense elements) x 4 B x 9 passes"}


def spawn_ranks(n, argv, script=None, timeout=None):
    """`python bench.py --gpus N` typed on its own (no torchrun around it): this process — which has not touched a GPU and never will
    — starts N rank processes through `python -m torch.distributed.run` (the launch line the driver itself uses), lets rank 0's JSON
    line through on stdout and returns a non-zero code when any rank failed.  Child processes, never an exec: a process that
    initialised the GPU must not be replaced, and the parent stays clean of HIP.  The reference's own multi-GPU entry starts its
    workers itself, too (nasrec/searcher/searcher.py:134-152)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL / cross-process device memory on this host driver
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()  # (the exact child this call started)
        out, _ = proc.communicate()
        sys.stderr.write("bench.py: the %d-rank run did not finish within %s s\n" % (n, timeout))
        return 124, None
    for ln in out.splitlines():
        at = ln.find('{"metric"')  # (other writers on the same pipe — gloo / RCCL banners from C++ — can land in front of it on one line)
        if at >= 0 and ln.rstrip().endswith("}"):
            line = ln[at:].rstrip()
            ln = ln[:at]
        if ln.strip():
            sys.stderr.write(ln + "\n")
    if proc.returncode != 0:
        sys.stderr.write("bench.py: a rank of the %d-rank run failed (exit code %d)\n" % (n, proc.returncode))
        return proc.returncode, line
    if line is None:
        sys.stderr.write("bench.py: the %d-rank run printed no result line\n" % n)
        return 1, None
    return 0, line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step's program instead of replaying a captured graph")
    ap.add_argument("--graph", action="store_true", help="replay the captured step even where launching its program is faster (default: the engine decides)")
    ap.add_argument("--steps-only", action="store_true",
                    help="profiler passes: only the set-up and the timed steps (no live roofline timing, forward-only / ablation legs or CPU baseline), "
                         "so that a rocprofv3 window holds nothing but real steps; the printed line then has no `roofline`")
    ap.add_argument("--force-dp-path", action="store_true", help="run the N>1 exchange code path on a single rank")
    ap.add_argument("--real-collectives", action="store_true",
                    help="with --force-dp-path on one rank: RCCL's all-gather / all-reduce kernels run inside the captured step (a one-rank gather is a copy kernel otherwise)")
    ap.add_argument("--ids", choices=["uniform", "zipf"], default="uniform",
                    help="id distribution (SURVEY 8d): uniform over each table = the cache-hostile primary case; zipf = Zipf(1.05) with 2 %% zeros")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: the config's batch PER GPU (global batch grows with N); strong: the config's batch is the GLOBAL batch")
    ap.add_argument("--table-sharding", choices=["none", "row"], default="none",
                    help="row: every rank owns a row range of every table (nasrec_amd/sharded_tables.py: ids / rows / row gradients by all-to-all, "
                         "owner-side dedup + clip + Adagrad) instead of replicated tables with an all-gather of row gradients")
    ap.add_argument("--id-pool", type=int, default=None, help="number of pre-generated id batches (default: enough to overflow the Infinity Cache)")
    args = ap.parse_args()
    w = WORKLOADS[args.config]
    fixed = w["mode"] == "fixed"
    steps = args.steps if args.steps is not None else (300 if fixed else 60)
    warmup = args.warmup if args.warmup is not None else (30 if fixed else 10)
    B = w["B"]
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        rc, line = spawn_ranks(args.gpus, sys.argv[1:])
        if line is not None and rc == 0:
            print(line)
        raise SystemExit(rc)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.scaling == "strong":
        if B % world_env:
            raise SystemExit("--scaling strong: the config's batch %d does not divide over %d GPUs" % (B, world_env))
        B = B // world_env  # per-GPU share of the config's (global) batch

    # stdout carries ONE line, the result: RCCL prints a version banner on stdout through C stdio (it came out BEHIND the JSON line
    # when stdout was a file), so everything else that is written to file descriptor 1 from here on — C libraries and Python's own
    # prints alike — goes to stderr, and the result line is written to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d, but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    import torch.distributed as dist
    # DRY RUN of the N-rank launch path on a box without GPUs (tests/test_bench_dry_gloo_cpu.py; never a measurement): the ranks form a
    # gloo group on the CPU and the engine behind DataParallelStep is a stand-in built by NASREC_BENCH_DRY_ENGINE = "file.py:factory"
    # (test infrastructure; bench.py itself provides none) — the rendezvous, the step loop, the fences, the max over ranks and the
    # result line are the ones the GPUs run
    dry = os.environ.get("NASREC_BENCH_DRY_ENGINE")
    if dry:
        device = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    if world > 1 or args.force_dp_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from nasrec_amd import _lib as L
    from nasrec_amd import plan as P
    from nasrec_amd.engine import SupernetEngine
    from nasrec_amd.parallel import DataParallelStep
    from nasrec_amd.search_space import ops_config_lib
    from nasrec_amd.utils.config import DATASETS
    from nasrec_amd.utils.lr_schedule import CosineAnnealingWarmupRestarts
    lib = L.load()

    ds = DATASETS[w["dataset"]]
    tables = [min(n, w["cap"]) if w.get("cap") else n for n in ds["tables"]]
    Fd, Fs = ds["Fd"], ds["Fs"]
    if dry:
        B = int(os.environ.get("NASREC_BENCH_DRY_BATCH", "4"))
        tables = [min(n, 50) for n in tables]
    n_pool = 4 if dry else args.id_pool if args.id_pool is not None else max(4, -(-(280 << 20) // (B * Fs * 64)))  # touched rows > 256 MiB per lap
    batches = BatchPool(n_pool, 16 if fixed else 4, B, Fd, tables, device, 1234 + rank, w["dataset"] == "avazu", args.ids)
    steps_per_epoch = w["train_limit"] // B
    sched = CosineAnnealingWarmupRestarts(steps_per_epoch, max_lr=LR_MAX if fixed else 0.12, min_lr=LR_MIN, warmup_steps=steps_per_epoch // 10)

    sharded = args.table_sharding == "row"

    class ShardedRun:
        """bench-side adapter: ShardedTableStep (row-sharded tables, f-4) behind the handful of attributes the reporting below reads"""
        exchange, graph = False, False

        def __init__(self, eng):
            from nasrec_amd.sharded_tables import EngineShardedOps, RowShardedTables, ShardedTableStep
            g = torch.Generator(device="cpu").manual_seed(7)
            self.t = RowShardedTables(tables, device, init_fn=lambda f, lo, hi: torch.randn(hi - lo, 16, generator=g) * (2.0 / (tables[f] + 16)) ** 0.5)
            self.ops = EngineShardedOps(eng, clip=5.0, eps=1e-2)
            self.stepper = ShardedTableStep(self.ops, self.t, B, clip=5.0, eps=1e-2, graph=not args.no_graph)
            self.graph = self.stepper.graph
            self._loss = None

        def step(self, int_x, cat_x, y, lr, choice=None):
            self._loss = self.stepper.step(int_x, cat_x, y, lr, choice=choice)

        def last_loss(self):
            return self._loss

        def last_plan(self):
            return self.ops.cp

    if fixed:
        choice_all = json.load(open(os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
        choice = {"macro": choice_all["macro"], "micro": choice_all["micro"]}
        # main_train.py:258-269: best-1shot sub-networks are built WITHOUT LayerNorm (use_layernorm hard-coded False)
        cfg = P.NetConfig(choice_all["num_blocks"], ops_config_lib[choice_all["config"]], False, "relu", fixed=True)
        if dry:
            import importlib.util
            path, fn = dry.rsplit(":", 1)
            spec = importlib.util.spec_from_file_location("_bench_dry_engine", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            eng = getattr(mod, fn)(choice_all, choice, Fd, Fs, tables)
        else:
            eng = SupernetEngine(cfg, Fd, Fs, tables, device=device, warm_choice=choice, world_size=world, host_embedding=sharded)
            eng.init_weights(seed=0)
        dp = ShardedRun(eng) if sharded else DataParallelStep(eng, choice, B, clip=5.0, eps=1e-2, graph=False if args.no_graph else True if args.graph else None,
                                                              force_exchange=args.force_dp_path, real_collectives=args.real_collectives)

        def one_step(i):
            bx = batches[i % len(batches)]
            dp.step(bx[0], bx[1], bx[2], sched.get_lr(), choice=choice)
            sched.step()
        parallelism = "dp%d (RCCL all-reduce of dense grads + all-gather of row-sparse embedding grads)" % world
    else:
        # train_supernet.py:241-257: weight-sharing supernet, full-path warm-up (train_utils.py:413-433), then the sampling strategy.
        # The path of every step comes from the global np.random stream (supernet.py:525-529): all ranks share the seed, so a
        # data-parallel step trains ONE path at the global batch — the reference's semantics at that batch size.
        from nasrec_amd.supernet.supernet import SuperNet
        torch.manual_seed(0)
        model = SuperNet(num_blocks=7, ops_config=ops_config_lib[w["space"]], use_layernorm=True, num_embeddings=tables, sparse_input_size=Fs,
                         path_sampling_strategy="full-path", fixed=False, anypath_choice="binomial-0.5",
                         table_sharding="row" if sharded else None).to(device)
        with torch.no_grad():
            model(batches[0][0][:64], batches[0][1][:64])
        eng = model._engine
        eng.init_weights(seed=0)
        eng.reserve(B, freeze_gc=True)  # plan slots sized for the largest path: no allocator call inside a step; one engine per process
        model.configure_path_sampling_strategy("default")
        np.random.seed(0)
        dp = DataParallelStep(eng, None, B, clip=5.0, eps=1e-2, graph=False, force_exchange=args.force_dp_path, real_collectives=args.real_collectives)
        if sharded:  # the module built the engine in host_embedding mode and adopted its nn.Embedding weights as this rank's shards
            class _ModuleRun(ShardedRun):
                def __init__(self):
                    from nasrec_amd.sharded_tables import ShardedTableStep
                    self.t, self.ops = model._sharded, model._sharded_ops
                    self.stepper = ShardedTableStep(self.ops, self.t, B, clip=5.0, eps=1e-2)
                    self._loss = None
            dp = _ModuleRun()

        seen_choices = []  # every sampled path of the run (the live roofline below averages the dominant kernel's algorithmic bytes over them)

        def one_step(i):
            bx = batches[i % len(batches)]
            ch = model._resolve_choice(None)  # the sampler of the drop-in module: global np.random, reference call order
            seen_choices.append(ch)
            dp.step(bx[0], bx[1], bx[2], sched.get_lr(), choice=ch)
            sched.step()
        parallelism = "dp%d (same sampled path on every rank; bucketed RCCL all-reduce of the path's dense grads overlapped with the " \
                      "backward + all-gather of row-sparse embedding grads)" % world

    if sharded:
        parallelism = "dp%d dense network (all-reduce of the path's dense grads) + row-sharded tables (all-to-all of ids / rows / row gradients, " \
                      "owner-side dedup + clip + Adagrad)" % world

    # measurement-only ablation (tools/dp_overhead.sh): leave one kind of collective out of the exchange step to see what it costs.  The
    # results of such a run are wrong by construction; the knob lives HERE, in the harness — nasrec_amd/parallel.py has no way to skip a collective
    skip = [k for k in os.environ.get("NASREC_BENCH_DP_SKIP", "").split(",") if k]
    if skip and getattr(dp, "exchange", False):
        from nasrec_amd.parallel import Collectives

        class Ablated(Collectives):
            def all_gather(self, out, local, async_op=False):
                return None if "gather" in skip else Collectives.all_gather(self, out, local, async_op)

            def all_reduce(self, t, async_op=False):
                return None if "reduce" in skip else Collectives.all_reduce(self, t, async_op)
        dp.coll = Ablated(args.real_collectives)

    def fence():
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            if device.type == "cuda":
                torch.cuda.synchronize(device)

    # Setup, not warm-up: the first step compiles the plan and captures the step graph; then the graph is replayed until the GPU has been
    # busy for ~0.1 s (clocks, TLBs and instruction caches of a box that sat idle through a minute of host-side setup), so that a short
    # timed region (the driver's --steps 20 --warmup 5 is 7 ms of GPU time) sees the steady state the default 300-step run sees.
    settle_ms = 0.0 if dry else float(os.environ.get("NASREC_BENCH_SETTLE_MS", "100"))
    one_step(0)
    fence()
    import gc
    gc.collect()
    gc.disable()  # (a launched — not replayed — step needs the host on time: no collector pause from here to the end of the timed region; a
    #                serving loop would do the same.  Before the settle phase: a collection between warm-up and timing idles the GPU for tens of ms)
    t_s = time.perf_counter()
    n_settle = 1
    while (time.perf_counter() - t_s) * 1e3 < settle_ms or n_settle < 4:
        one_step(n_settle)
        n_settle += 1
        if n_settle % 16 == 0 and device.type == "cuda":
            torch.cuda.synchronize(device)
    fence()
    for i in range(warmup):
        one_step(n_settle + i)
    fence()
    # the timed region: exactly `steps` steps between two fences, nothing else in the stream (an event record per step is a barrier packet
    # with a signal: between launched kernels it opens a ~10 us gap every few steps, rocprofv3 kernel trace)
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(n_settle + warmup + i)
    fence()
    dt = time.perf_counter() - t0
    # per-step durations (median_ms_per_step, the step-level roofline) from a second, untimed pass with an event behind every step
    nper = 0 if dry else min(steps, 300)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nper + 1)] if nper else []
    if nper:
        ev[0].record()
    for i in range(nper):
        one_step(n_settle + warmup + steps + i)
        ev[i + 1].record()
    fence()
    gc.enable()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    per_step = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(nper)]) if nper else np.array([dt / steps * 1e3])
    loss = float(dp.last_loss().item())
    if not (loss == loss):
        raise SystemExit("loss is NaN")

    cpx = getattr(dp, "cp", None)
    cpx = getattr(cpx, "cp", cpx)  # (the exchange step's DPPlan wraps the engine's plan)
    result = {
        "metric": "supernet samples/sec at batch 256 (Criteo-shape), 1/2/4/8 MI355X",
        "value": B * world * steps / dt, "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup, "setup_steps": n_settle,
        "ms_per_step": dt / steps * 1e3, "median_ms_per_step": float(np.median(per_step)), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": w["name"], "baseline_config": args.config, "per_gpu_batch": B, "global_batch": B * world, "graph": dp.graph,
                   "step_submission": ("one graph replay per step" if dp.graph else "the step's program launched by one host call per step (nasrec_program_run)") if fixed else "programs launched per sampled path",
                   "parallelism": parallelism,
                   "ids": "%s over each table; %d pre-generated id batches = %.0f MB of distinct-ish table rows per lap of the pool (Infinity Cache: 256 MB)"
                          % (args.ids, len(batches), len(batches) * B * Fs * 64 / 1e6),
                   "embedding_update": "row-sparse clip+Adagrad (== dense reference update for weight_decay 0)", "table_sharding": args.table_sharding,
                   # round 6: where the step's buffers live and which form the joint forward + backward program takes (engine.py: _UC_ARENA, persist)
                   "plan_buffers": ("uncached device memory (hipDeviceMallocUncached)" if (fixed and getattr(getattr(cpx, "arena", None), "uncached", False))
                                    else "torch allocator"),
                   "joint_program": ("persistent launch (in-kernel dependencies)" if getattr(cpx, "persistent", False) else
                                     "one launch per dependency level") if fixed else "one launch per operator",
                   # the level schedule the plan's compile-time tuner kept (engine._tune_levels): outside the timed region, same bits whatever it picks
                   "level_schedule_tuning": getattr(cpx, "level_tuning", None)},
        "final_loss": loss,
    }
    if dry:
        result["data"] = "synthetic; DRY RUN of the launch path on the CPU (gloo, stand-in engine, batch %d, 50-row tables): NOT a measurement" % B
        result["config"]["dry_run"] = True
    if dp.exchange:
        pl = dp._last[1]
        result["config"]["dp_exchange"] = {
            "backend": dist.get_backend() + (" (= RCCL on ROCm)" if dist.get_backend() == "nccl" else ""), "world_size": dist.get_world_size(),
            "captured_in_one_graph": bool(getattr(pl, "step_graph", None)),
            "launches_up_to_cut": getattr(pl, "cuts", None),
            "pieces": [{"allreduce_ranges": len(list(rg)), "allreduce_MB": sum(n for _, n in rg) * 4 / 1e6} for _, rg in pl.segments],
            "allgather_MB": {"ids": B * world * Fs * 8 / 1e6, "row_gradients": B * world * Fs * 64 / 1e6},
            # the last pieces' dense gradients ride behind the rows in the row-gradient all-gather (no all-reduce of their own)
            "packed_tail_floats_per_rank": dp.tail_n, "first_packed_piece": getattr(pl, "first_packed", None),
            "real_collectives_on_one_rank": bool(args.real_collectives),
            # where the id-only half of the global-batch row dedup runs (parallel._exchange_step): "main" = on the compute stream in front of
            # the optimizer (exposed; the default: a fork to a second stream inside a captured graph costs more than the launch),
            # "side" / "chain" = beside the forward on a second stream; None = the one-launch dedup kernels (no separate id half)
            "id_half_split_from_the_row_sums": dp.ids_half is not None,
            "id_half_mode": (__import__("nasrec_amd.parallel", fromlist=["_IDS_MODE"])._IDS_MODE if getattr(dp, "_side", None) is not None else "main") if dp.ids_half is not None else None}

    if rank == 0 and not args.steps_only and not dry:  # (N > 1: the roofline of the dominant launch and the step accounting are rank 0's; forward-only and the CPU baseline are N = 1 legs)
        sp = torch.cuda.current_stream(device).cuda_stream
        cp = dp.last_plan()
        allg = [d for d in P.iter_ops(cp.fwd.descs + cp.bwd.descs) if isinstance(d, L.GemmDesc)]
        if fixed and world == 1 and not sharded:
            # ---- forward-only throughput (eval path) ---------------------------------------------------------------
            fgraph = False if args.no_graph else True if args.graph else eng.prefers_graph(B)  # (replay or launch: engine.prefers_graph)
            eng.compile(choice, B, train=False, graph=fgraph)
            for _ in range(20):
                eng.forward(batches[0][0], batches[0][1], graph=fgraph)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for i in range(200):
                eng.forward(batches[i][0], batches[i][1], graph=fgraph)
            torch.cuda.synchronize(device)
            result["forward_only_samples_per_s"] = B * 200 / (time.perf_counter() - t1)

        if fixed and world == 1 and not args.force_dp_path and not sharded:
            # ---- secondary figure: the same step with the forward operators nobody reads dropped (opt-in, never `value`): the
            # reference computes block 5 of this architecture and throws it away (last_n_blocks_out = 1, no later block selects it)
            eng.dead_code_elimination = True
            eng._plans.clear()
            eng._last_plan = None
            dp2 = DataParallelStep(eng, choice, B, clip=5.0, eps=1e-2, graph=False if args.no_graph else True if args.graph else None)
            for i in range(20):
                dp2.step(*batches[i], sched.get_lr())
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for i in range(200):
                dp2.step(*batches[20 + i], sched.get_lr())
            torch.cuda.synchronize(device)
            dt2 = (time.perf_counter() - t1) / 200
            cp2 = dp2.last_plan()
            dead = [d for d in P.iter_ops(cp2.dead_forward)]
            result["with_dead_forward_ops_dropped"] = {
                "samples_per_s": B / dt2, "ms_per_step": dt2 * 1e3, "dropped_launches": len(dead),
                "dropped_gemm_gflop": sum(gemm_flops(d) for d in dead if isinstance(d, L.GemmDesc)) / 1e9,
                "note": "opt-in (NASREC_DCE=1): identical logits / loss / gradients / parameters; the headline `value` executes every operator the reference executes"}
            eng.dead_code_elimination = False
            eng._plans.clear()
            eng._last_plan = None
            cp = dp.cp = eng.compile(choice, B, True, 5.0, 1e-2, graph=dp.graph)

        if fixed and world == 1 and not args.force_dp_path and not sharded and os.environ.get("NASREC_BENCH_HARNESS_ROUTE", "1") != "0":
            # ---- secondary figure: the reference's harness step UNCHANGED on the drop-in module (dense table gradients, torch's clip and
            # Adagrad over the full tables) — what `main_train.py` costs when it does not take the fused engine step (never `value`)
            try:
                result["unchanged_harness_route"] = unchanged_harness_route(device, B, tables, Fd, Fs, choice_all, [batches[i] for i in range(8)])
            except Exception as e:  # noqa: BLE001  (a secondary leg must not take the line down)
                result["unchanged_harness_route"] = {"error": repr(e)[:300]}

        if world == 1 and not sharded:
            # ---- embedding stem on COLD rows: the staging launch (batch copy + gather of B x Fs 64-byte rows, a1 of SURVEY 8a) over
            # distinct id batches of the pool — random 64-B rows out of HBM, HBM-latency bound -----------------------------------
            ng = min(200, len(batches))
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(5):
                eng._stage_inputs(sp, cp, *batches[len(batches) - 1 - i], 1e-3)
            ea.record()
            for i in range(ng):
                eng._stage_inputs(sp, cp, *batches[i], 1e-3)
            eb.record()
            torch.cuda.synchronize(device)
            us = ea.elapsed_time(eb) / ng * 1e3
            result["embedding_gather_cold"] = {"avg_launch_us": us, "rows": B * Fs, "bytes": B * Fs * 128,
                                               "achieved_GBps": B * Fs * 128 / us / 1e3, "bound": "hbm latency (random 64-B rows: read + write)",
                                               "frac_of_hbm_peak": B * Fs * 128 / us / 1e3 / HBM_PEAK_GBS}

        # ---- the largest GEMM launch of the (last) step's plan, fp32 MFMA bound ---------------------------------------------
        allg = [d for d in P.iter_ops(cp.fwd.descs + cp.bwd.descs) if isinstance(d, L.GemmDesc)]  # (of the CURRENT plan: the leg above re-compiled it)
        dom = max(allg, key=gemm_flops)
        ms = time_desc(lib, L, sp, dom, iters=200 if fixed else 30)
        fl = gemm_flops(dom)
        s0 = dom.seg[0]
        big = {"bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None,
               "kernel": "%s<%d,%d,%d> M=%d N=%d K=%d nseg=%d splitk=%d" % (
                   P.gemm_kernel_name(dom), dom.amode, dom.bmode, dom.cmode, s0.M, s0.N, sum(dom.seg[q].K for q in range(dom.nseg)),
                   dom.nseg, dom.splitk),
               "flops_per_launch": fl, "avg_launch_us": ms * 1e3,
               # operands + outputs of this launch, each once (split-K slabs written and read back included)
               "algorithmic_bytes": gemm_algorithmic_bytes(dom)}
        # HBM/fabric traffic of that launch: PMC counters cannot be collected from inside this process; a measured value is
        # quoted only when profiles/ holds one taken from THIS build (csrc hash) for exactly this launch, else null
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "dominant_gemm_traffic_cfg%d.json" % args.config)))
            if tj.get("build_id") == csrc_build_id() and tj.get("kernel") == big["kernel"]:
                big["traffic"] = tj["traffic_bytes_per_launch"]
                big["traffic_unit"] = "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, %s)" % tj.get("source", "profiles/")
                big["traffic_source"] = "profiles/dominant_gemm_traffic_cfg%d.json (%s; not this run)" % (args.config, tj.get("source", "rocprofv3 --pmc"))
                big["algorithmic_bytes"] = tj["algorithmic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        result["roofline_largest_gemm"] = big
        # ---- `roofline`: the kernel that DOMINATES THE STEP BY TIME.  Every launch of the step is timed on its own; launches are
        # grouped by device kernel; the kernel with the largest share of the step's time is the line's `roofline`, with the aggregate
        # rate of its launches (sum of algorithmic flops / sum of launch durations) against the fp32 MFMA roof, and the same for
        # its algorithmic bytes against the HBM roof.  `roofline_levels` is the table it is computed from. -------------------------
        from nasrec_amd import schedule as S
        step_descs = [cp.stage] + (list(cp.fb.descs) if getattr(cp, "fb", None) is not None else list(cp.fwd.descs) + list(cp.bwd.descs)) + list(cp.opt.descs)
        if not dp.exchange or fixed:
            n_dense = dense_floats_walked(eng, cp)  # (what the optimizer's launches walk: the plan's chunk table)
            # (the table re-launches the optimizer's descriptors a hundred times: what they change — dense parameters and state, the touched
            # rows and their state, the row gradients that are summed in place — is put back afterwards)
            with torch.no_grad():
                ids_now = cp.cat_x.clone()
                snap = [eng.flat_p.clone(), eng.flat_s.clone(), cp.sparse0.grad_tensor().clone() if not dp.exchange else None,
                        [t[ids_now[:, f]].clone() for f, t in enumerate(eng.tables)], [t[ids_now[:, f]].clone() for f, t in enumerate(eng.table_state)]]
            rows = launch_table(lib, L, P, S, sp, step_descs, 100 if fixed else 10, n_dense)
            # ... and every launch once more IN THE STEP'S ORDER (the durations the roofline is built from: a back-to-back relaunch of one
            # descriptor runs 5 - 30 % faster than the same launch inside the step, whose operands were just written by other kernels)
            blk = torch.empty(4096, 4096, device=device).normal_()
            blk_out = torch.empty_like(blk)

            def blocker():
                for _ in range(4 if fixed else 6):
                    torch.mm(blk, blk, out=blk_out)

            def warm_step():
                for dsc in step_descs:
                    L.check(lib.nasrec_launch(sp, C.addressof(getattr(dsc, "launch_as", dsc))))
            us_in, us_raw, seq_us = launch_table_in_step(lib, L, sp, step_descs, 30 if fixed else 5, blocker, warm_step)
            for r, u, w_ in zip(rows, us_in, us_raw):
                r["us_isolated"], r["us"], r["us_event_interval"] = r["us"], float(u), float(w_)
            result["roofline_in_step_timing"] = {"launches": len(step_descs), "sum_event_intervals_us": float(us_raw.sum()), "plain_sequence_us": seq_us,
                                                 "event_packet_excess_us_per_launch": (float(us_raw.sum()) - seq_us) / len(step_descs)}
            del blk, blk_out
            with torch.no_grad():
                eng.flat_p.copy_(snap[0])
                eng.flat_s.copy_(snap[1])
                if snap[2] is not None:
                    cp.sparse0.grad_tensor().copy_(snap[2])
                for f in range(len(eng.tables)):
                    eng.tables[f][ids_now[:, f]] = snap[3][f]
                    eng.table_state[f][ids_now[:, f]] = snap[4][f]
            if fixed:  # (a sampled supernet path has ~150 launches: its table stays in roofline_kernels, aggregated by kernel)
                result["roofline_levels"] = rows
            agg = {}
            for r in rows:
                a = agg.setdefault(r["kernel"], {"us": 0.0, "us_isolated": 0.0, "mflop": 0.0, "alg_KB": 0.0, "launches": 0})
                a["us"] += r["us"]
                a["us_isolated"] += r["us_isolated"]
                a["mflop"] += r["mflop"]
                a["alg_KB"] += r["alg_KB"]
                a["launches"] += 1
            tot_us = sum(a["us"] for a in agg.values())
            # (the two instantiations of the worklist kernel are one kernel for this purpose: same bodies, the <true> variant adds the
            # Transformer backward and its LDS)
            fam = {}
            for k, a in agg.items():
                f = fam.setdefault(k.split("<")[0], {"us": 0.0, "us_isolated": 0.0, "mflop": 0.0, "alg_KB": 0.0, "launches": 0, "variants": []})
                for key in ("us", "us_isolated", "mflop", "alg_KB", "launches"):
                    f[key] += a[key]
                f["variants"].append(k)
            name, a = max(fam.items(), key=lambda kv: kv[1]["us"])
            tf = a["mflop"] / a["us"] if a["us"] else 0.0      # MFLOP / us = TFLOP/s
            gbs = a["alg_KB"] / a["us"] if a["us"] else 0.0    # KB / us = GB/s
            mf, hf = tf / MFMA_F32_PEAK_TFLOPS, gbs / HBM_PEAK_GBS
            hbm = hf > mf  # the roof the kernel is closer to
            result["roofline"] = {
                "bound": "hbm" if hbm else "mfma", "achieved": gbs if hbm else tf, "peak": HBM_PEAK_GBS if hbm else MFMA_F32_PEAK_TFLOPS,
                "unit": "GB/s" if hbm else "TFLOP/s", "frac": hf if hbm else mf,
                "traffic": None, "kernel": name, "variants": sorted(a["variants"]), "launches_per_step": a["launches"],
                "share_of_step_time": a["us"] / tot_us if tot_us else None,
                "flops_per_launch": a["mflop"] * 1e6 / a["launches"], "avg_launch_us": a["us"] / a["launches"],
                "avg_launch_us_isolated": a["us_isolated"] / a["launches"],
                "duration_source": "HIP events between consecutive launches of real steps, in the step's own order (mean of %d steps), each interval reduced by the "
                                   "mean excess the event packets add (roofline_in_step_timing: the corrected durations sum to the launches' back-to-back time); "
                                   "`avg_launch_us_isolated` = back-to-back relaunches of each descriptor" % (30 if fixed else 5),
                "algorithmic_bytes": a["alg_KB"] * 1e3 / a["launches"],
                "mfma_achieved_TFLOPs": tf, "mfma_frac": mf, "hbm_achieved_GBps": gbs, "hbm_frac": hf,
                "note": "aggregate over this kernel's launches in one step (sum of flops / sum of in-step launch durations); a launch of this "
                        "kernel is one dependency level of the step, latency-bound at batch 256 (DESIGN.md 3); per-launch rows in roofline_levels"}
            result["roofline_kernels"] = {k: dict(v, share=v["us"] / tot_us) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"])}
            if not fixed and not dp.exchange:
                # a supernet step runs a different sampled path every time: `traffic` (a counter mean over the launches of MANY paths) is only
                # comparable with algorithmic bytes averaged over many paths too — the launches of the last path alone can be twice or half that
                tb, tn = 0.0, 0
                for chs in seen_choices[-(steps + warmup):]:
                    cpi = eng.compile(chs, B, True, 5.0, 1e-2, graph=False)
                    for dsc in list(cpi.fwd.descs) + list(cpi.bwd.descs) + list(cpi.opt.descs):
                        if launch_kernel_name(L, P, dsc).split("<")[0] == name:
                            tb += launch_work(L, S, dsc, n_dense)[1]
                            tn += 1
                if tn:
                    result["roofline"]["algorithmic_bytes_last_path"] = result["roofline"]["algorithmic_bytes"]
                    result["roofline"]["algorithmic_bytes"] = tb / tn
                    result["roofline"]["algorithmic_bytes_note"] = "mean per launch over the last %d sampled paths of the run (%d launches of this kernel)" % (len(seen_choices[-(steps + warmup):]), tn)
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "dominant_kernel_traffic_cfg%d.json" % args.config)))
                if tj.get("build_id") == csrc_build_id() and tj.get("kernel") == name:
                    result["roofline"]["traffic"] = tj["traffic_bytes_per_launch"]
                    result["roofline"]["traffic_unit"] = "bytes per launch, mean over the kernel's launches (PMC FETCH_SIZE x2 + WRITE_SIZE, %s)" % tj.get("source", "profiles/")
                    # the counters cannot be read from inside this process: the figure is a committed artefact of a rocprofv3 --pmc run of this
                    # same command on this same build (csrc hash checked above), NOT a measurement of this run
                    result["roofline"]["traffic_source"] = "profiles/dominant_kernel_traffic_cfg%d.json (%s; not this run)" % (args.config, tj.get("source", "rocprofv3 --pmc"))
                else:
                    result["roofline"]["traffic_source"] = "none for this build (profiles/dominant_kernel_traffic_cfg%d.json is from build %s)" % (args.config, tj.get("build_id"))
            except (OSError, KeyError, ValueError):
                pass
        else:
            result["roofline"] = big
        # ---- whole-step accounting -----------------------------------------------------------------------------
        step_s = float(np.median(per_step)) * 1e-3
        step_flops = sum(gemm_flops(d) for d in allg)
        result["step_gemm_gflop"] = step_flops / 1e9
        result["step_mfma_frac"] = step_flops / step_s / 1e12 / MFMA_F32_PEAK_TFLOPS
        # byte model of the step under the ACTIVE (row-sparse) embedding semantics: dense parameters x 4 B x 9 passes (grad write,
        # norm read, Adagrad grad read + state r/w + param r/w, forward read, backward read) + touched rows x 64 B x 7
        n_used = sum(int(np.prod(eng.shapes[n])) for n in cp.used_params if not n.startswith("_embedding."))
        step_bytes = n_used * 4 * 9 + B * Fs * 64 * 7
        result["roofline_step"] = {"bound": "hbm", "achieved": step_bytes / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": step_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                                   "bytes_model": "(%d dense parameters on the path x 4 B x 9 passes) + (B x Fs touched rows x 64 B x 7) = %.1f MB per step, "
                                                  "row-sparse embedding semantics" % (n_used, step_bytes / 1e6),
                                   "launches_per_step": len(cp.fwd.descs) + len(cp.bwd.descs) + len(cp.opt.descs) + 1,
                                   "launch_floor_ms": (len(cp.fwd.descs) + len(cp.bwd.descs) + len(cp.opt.descs) + 1) * 1.65e-3}
        if not args.no_cpu_baseline and world == 1:
            # torch CPU ops stop scaling (and start thrashing) far below this box's logical core count: use one
            # socket's worth of threads at most, and state the number
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            result["cpu_baseline"] = cpu_baseline(w, choice if fixed else None, tables, int(os.environ.get("NASREC_CPU_THREADS", min(avail, 32))), Fd)
            result["cpu_baseline"]["host_logical_cpus"] = avail
            result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
    if rank == 0:
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
