#!/usr/bin/env python3
"""Gaps between consecutive kernels of the cfg-2 step from a rocprofv3 kernel trace (tools/launch_gaps.py DIR): behind the eager staging
launch (-> the step graph's first kernel), in front of it (graph's last kernel -> next step's staging launch), and inside the graph."""
import csv, glob, statistics, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
after, before, inside = [], [], []
for a, b in zip(rows[:-1], rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    (after if "stage_inputs" in a["Kernel_Name"] else before if "stage_inputs" in b["Kernel_Name"] else inside).append(g)
def st(v):
    v = sorted(x for x in v if x < 200)
    return "n=%d median %.2f us, 10%% %.2f, 90%% %.2f" % (len(v), statistics.median(v), v[len(v) // 10], v[-len(v) // 10]) if v else "-"
print("staging launch -> first kernel of the step graph:", st(after))
print("last kernel of the step graph -> next staging launch:", st(before))
print("between kernels inside the step graph:", st(inside))
stage = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "stage_inputs" in r["Kernel_Name"]]
print("staging kernel itself:", st(stage))
