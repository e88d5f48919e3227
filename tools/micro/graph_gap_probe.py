#!/usr/bin/env python3
"""What does the boundary between a replayed graph and the next launch cost?  Three small kernels captured as a graph (torch.cuda.CUDAGraph);
(a) replays back to back, (b) an eager kernel between replays (the step's staging launch).  Run under rocprofv3 --kernel-trace and read the
gaps with MODE=read DIR=..."""
import os, sys, csv, glob, statistics
if os.environ.get("MODE") == "read":
    f = sorted(glob.glob(os.environ["DIR"] + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    gaps = {}
    for a, b in zip(rows[:-1], rows[1:]):
        ka = "mul" if "mul" in a["Kernel_Name"].lower() else "add" if "add" in a["Kernel_Name"].lower() else "other"
        kb = "mul" if "mul" in b["Kernel_Name"].lower() else "add" if "add" in b["Kernel_Name"].lower() else "other"
        gaps.setdefault((ka, kb), []).append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    for k, v in sorted(gaps.items()):
        v = sorted(x for x in v if x < 500)
        if len(v) >= 20:
            print("%s -> %s: n=%d median gap %.2f us (10%% %.2f, 90%% %.2f)" % (k[0], k[1], len(v), statistics.median(v), v[len(v) // 10], v[-len(v) // 10]))
    sys.exit(0)
import torch
x = torch.zeros(1 << 16, device="cuda")
y = torch.zeros(1 << 16, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        x.add_(1.0); x.add_(1.0); x.add_(1.0)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        x.add_(1.0); x.add_(1.0); x.add_(1.0)   # "add" kernels: the graph
    for _ in range(200):   # (a) graph, graph, ...
        g.replay()
    s.synchronize()
    for _ in range(200):   # (b) eager "mul" kernel between replays
        y.mul_(1.0)
        g.replay()
    s.synchronize()
