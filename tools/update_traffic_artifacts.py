#!/usr/bin/env python3
"""profiles/pmc_tmp/cfg{2,3}_{FETCH_SIZE,WRITE_SIZE}.csv (rocprofv3 --pmc counter_collection.csv of the four passes described in
profiles/README.md) -> profiles/dominant_gemm_traffic_cfg{2,3}.json with the csrc hash of the current build + per-kernel summaries."""
import csv, hashlib, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha1()
d = os.path.join(ROOT, "nasrec_amd", "csrc")
for f in sorted(os.listdir(d)):
    h.update(open(os.path.join(d, f), "rb").read())
bid = h.hexdigest()[:16]
T = os.path.join(ROOT, "profiles", "pmc_tmp")
res = {}
for nm in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(os.path.join(T, "cfg2_%s.csv" % nm))))
    v = [float(r["Counter_Value"]) for r in rows if "gemm_ring_kernel<0, 0, 0, 1024, 64, 64, 64, false>" in r["Kernel_Name"] and r["Grid_Size"] == "245760"]
    if nm == "FETCH_SIZE":
        v = [x for x in v if x > 8000]  # the K = 1565 launch (the K = 780 launch of the same kernel fetches half)
    e = [float(r["Counter_Value"]) for r in rows if "gemm_splitk_epilogue" in r["Kernel_Name"] and r["Grid_Size"] == "196608"]
    res[nm] = (statistics.mean(v), statistics.mean(e), len(v), len(e))
p2 = os.path.join(ROOT, "profiles", "dominant_gemm_traffic_cfg2.json")
j = json.load(open(p2))
fk, wk = res["FETCH_SIZE"][0] + res["FETCH_SIZE"][1], res["WRITE_SIZE"][0] + res["WRITE_SIZE"][1]
j["build_id"] = bid
j["fetch_size_kb_raw"] = {"main": res["FETCH_SIZE"][0], "second_pass": res["FETCH_SIZE"][1]}
j["write_size_kb"] = {"main": res["WRITE_SIZE"][0], "second_pass": res["WRITE_SIZE"][1]}
j["traffic_bytes_per_launch"] = int(round((2 * fk + wk) * 1024))
json.dump(j, open(p2, "w"), indent=1)
res3 = {}
for nm in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(os.path.join(T, "cfg3_%s.csv" % nm))))
    v = [float(r["Counter_Value"]) for r in rows if "gemm_fast_kernel<0, 0, false>" in r["Kernel_Name"] and r["Grid_Size"] == "335872"]
    res3[nm] = (statistics.mean(v), len(v))
p3 = os.path.join(ROOT, "profiles", "dominant_gemm_traffic_cfg3.json")
j3 = json.load(open(p3))
j3["build_id"] = bid
j3["fetch_size_kb_raw"], j3["write_size_kb"] = res3["FETCH_SIZE"][0], res3["WRITE_SIZE"][0]
j3["traffic_bytes_per_launch"] = int(round((2 * res3["FETCH_SIZE"][0] + res3["WRITE_SIZE"][0]) * 1024))
json.dump(j3, open(p3, "w"), indent=1)
for c in (2, 3):
    for nm, short in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), os.path.join(T, "cfg%d_%s.csv" % (c, nm))], capture_output=True, text=True).stdout
        open(os.path.join(ROOT, "profiles", "r02_bench_cfg%d_pmc_%s.csv" % (c, short)), "w").write(out)
print(bid, res, res3, j["traffic_bytes_per_launch"], j3["traffic_bytes_per_launch"])
