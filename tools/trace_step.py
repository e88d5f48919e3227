#!/usr/bin/env python3
"""Summarise one steady-state training step from a rocprofv3 --kernel-trace CSV: kernel sequence, durations, gaps."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# step boundaries: the last kernel of the optimizer tail
idx = [i for i, n in enumerate(names) if n.startswith("adagrad_rows") or n.startswith("opt_apply")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[which - 1] + 1, idx[which] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
tot_k = 0
agg = collections.OrderedDict()
prev_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = e - s
    tot_k += d
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
    g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "")
    wg = (r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "")
    if len(sys.argv) > 3:
        print("%8.1f us  +%6.1f gap  %7.1f us  %-40s grid=%s wg=%s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, d / 1e3, nm[:40], g, wg))
    prev_end = e
    k = agg.setdefault(nm, [0, 0])
    k[0] += 1
    k[1] += d
wall = int(step[-1]["End_Timestamp"]) - t0
print("step %d: %d kernels, wall %.1f us, kernel-busy %.1f us (%.0f%%)" % (which, len(step), wall / 1e3, tot_k / 1e3, 100.0 * tot_k / wall))
for nm, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s x%-3d %8.1f us  (%.1f%%)" % (nm[:44], c, d / 1e3, 100.0 * d / tot_k))
