#!/usr/bin/env python3
"""Host-side cost of a supernet step (plan compile per sampled path + enqueue) against its GPU time: the step is asynchronous, so
the host may run ahead and compile step t+1 while step t executes; kernel-time / wall >= 0.97 is the bar."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.supernet.supernet import SuperNet
from nasrec_amd.utils.config import DATASETS
cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = bench.WORKLOADS[cfgid]
ds = DATASETS[w["dataset"]]
tables = [min(n, w["cap"]) if w.get("cap") else n for n in ds["tables"]]
B = w["B"]
dev = torch.device("cuda", 0)
batches = bench.synthetic_batches(2, B, ds["Fd"], tables, dev, 1, zero_dense=(w["dataset"] == "avazu"))
torch.manual_seed(0)
model = SuperNet(num_blocks=7, ops_config=ops_config_lib[w["space"]], use_layernorm=True, num_embeddings=tables, sparse_input_size=ds["Fs"],
                 path_sampling_strategy="full-path", fixed=False, anypath_choice="binomial-0.5").to(dev)
with torch.no_grad():
    model(batches[0][0][:64], batches[0][1][:64])
model._engine.init_weights(0)
model._engine.reserve(B, freeze_gc=True)
model.configure_path_sampling_strategy("default")
np.random.seed(0)
for i in range(8):
    model.engine_train_step(*batches[i % 2], lr=1e-3)
torch.cuda.synchronize()
N = 40
eng = model._engine
host, comp, ev = [], [], [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(N):
    h0 = time.perf_counter()
    # the plan compile of this step's path, timed on its own (it is pure host work; the enqueue behind it can block on a full
    # hardware queue when the host runs several steps ahead, which is back-pressure, not cost)
    st = np.random.get_state()
    ch = model._resolve_choice(None)
    eng.compile(ch, B, True, 5.0, 1e-2, graph=False)
    h1 = time.perf_counter()
    np.random.set_state(st)  # engine_train_step draws the same path again and hits the plan just compiled
    model.engine_train_step(*batches[i % 2], lr=1e-3)
    host.append(time.perf_counter() - h0)
    comp.append(h1 - h0)
    ev[i + 1].record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
gpu = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(N)])
print("cfg %d: wall %.2f ms/step; path sampling + plan compile %.2f ms/step mean, %.2f median, %.2f max; whole host call (incl. enqueue, which blocks when the "
      "host is several steps ahead) %.2f mean %.2f median %.2f max; GPU span per step median %.2f ms; sum of GPU spans / wall = %.3f" % (
    cfgid, wall / N * 1e3, np.mean(comp) * 1e3, np.median(comp) * 1e3, np.max(comp) * 1e3, np.mean(host) * 1e3, np.median(host) * 1e3, np.max(host) * 1e3,
    np.median(gpu), gpu.sum() / (wall * 1e3)))
# the same host call with an EMPTY queue in front of it (a synchronise before every step: no back-pressure from the launch queue in the
# numbers) — what the host itself costs per step, enqueue included
host2 = []
for i in range(N):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    model.engine_train_step(*batches[i % 2], lr=1e-3)
    host2.append(time.perf_counter() - h0)
torch.cuda.synchronize()
print("cfg %d, empty queue before every step: whole host call %.2f ms mean, %.2f median, %.2f max (compiled host modules: %s)" % (
    cfgid, np.mean(host2) * 1e3, np.median(host2) * 1e3, np.max(host2) * 1e3, __import__("nasrec_amd").host_modules_compiled()))
