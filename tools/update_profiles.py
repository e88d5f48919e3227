#!/usr/bin/env python3
"""Post-processing of tools/collect_profiles.sh (one script for every round: NASREC_PROFILE_TAG / --tag names the round, default r06).
  --summarise DIR   (on the GPU box) reduce the rocprofv3 outputs under DIR to small files in DIR/summary: kernel stats, per-kernel counter
                    means, and the per-dispatch durations / FETCH_SIZE / WRITE_SIZE of the dominant launch of cfg 2 and cfg 3
  --install DIR     (in the repo) copy DIR/summary and the result lines into profiles/<tag>_*, write profiles/dominant_gemm_traffic_cfg{2,3}.json
                    with the csrc hash of the current build (bench.py quotes roofline.traffic only when hash and kernel string match)"""
import collections, csv, glob, hashlib, json, os, shutil, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("NASREC_PROFILE_TAG", "r06")
if "--tag" in sys.argv:
    TAG = sys.argv[sys.argv.index("--tag") + 1]
CFGS = (2, 3, 4, 5)


def build_id():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "nasrec_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    return name.split("(")[0].replace("void ", "")


def one(pattern):
    g = sorted(glob.glob(pattern, recursive=True))
    return g[-1] if g else None


def split_upper(vals):
    """the dominant launch shares its kernel name with a smaller product: keep the upper cluster"""
    if len(vals) < 4:
        return vals
    s = sorted(vals)
    lo, hi = s[len(s) // 10], s[-1 - len(s) // 10]
    if hi < 1.3 * lo:
        return vals
    thr = 0.5 * (lo + hi)
    return [v for v in vals if v > thr]


def trace_name(bench_kernel):
    """bench.py's `roofline.kernel` string ('gemm_fast_kernel<0,0,0> M=...') -> the kernel name as rocprofv3 prints it"""
    head = bench_kernel.split(" ")[0]
    if head.startswith("gemm_fast_kernel<"):
        a, b, o = head[len("gemm_fast_kernel<"):-1].split(",")
        return "gemm_fast_kernel<%s, %s, %s>" % (a, b, "true" if o != "0" else "false")
    return head.split("<")[0]


def dominant_group(trace_csv, name_part):
    rows = [r for r in csv.DictReader(open(trace_csv)) if name_part in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        grid = r.get("Grid_Size") or str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        by[(short(r["Kernel_Name"]), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    by = {k: v for k, v in by.items() if len(v) >= 20}
    key = max(by, key=lambda k: statistics.median(by[k]))
    return key, by[key]


def summarise(O):
    S = os.path.join(O, "summary")
    os.makedirs(S, exist_ok=True)
    out = {}
    for c in CFGS:
        res = os.path.join(O, "bench_cfg%d_result.json" % c)
        if not os.path.exists(res):
            continue
        line = json.loads(open(res).read().strip().splitlines()[-1])
        part = trace_name(line.get("roofline_largest_gemm", line["roofline"])["kernel"])
        family = line["roofline"]["kernel"].split("<")[0].split(" ")[0]  # the time-dominant kernel (all its instantiations and grids)
        ks = one(os.path.join(O, "trace_cfg%d" % c, "**", "*kernel_stats.csv"))
        kt = one(os.path.join(O, "trace_cfg%d" % c, "**", "*kernel_trace.csv"))
        if ks:
            shutil.copy(ks, os.path.join(S, "bench_cfg%d_kernel_stats.csv" % c))
        if not kt:
            continue
        (name, grid), dur = dominant_group(kt, part)
        dur = split_upper(dur)
        rec = {"kernel": name, "grid_size": grid, "dispatches": len(dur), "mean_us": statistics.mean(dur), "median_us": statistics.median(dur),
               "min_us": min(dur), "max_us": max(dur), "durations_us": [round(x, 2) for x in dur]}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            # per-kernel means: the window of real steps (`--steps-only`); the rows of the LARGEST GEMM launch: the window that also holds
            # the live roofline timing (that launch thirty times over — a sampled path's largest launch hardly repeats otherwise)
            cc = one(os.path.join(O, "pmc_cfg%d_%s" % (c, ctr), "**", "*counter_collection.csv"))
            ccf = one(os.path.join(O, "pmcfull_cfg%d_%s" % (c, ctr), "**", "*counter_collection.csv")) or cc
            if not cc:
                continue
            rows = list(csv.DictReader(open(cc)))
            rows_full = rows if ccf == cc else list(csv.DictReader(open(ccf)))
            agg = collections.defaultdict(list)
            for r in rows:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            with open(os.path.join(S, "bench_cfg%d_pmc_%s.csv" % (c, ctr.split("_")[0].lower())), "w") as f:
                f.write("kernel,counter,dispatches,mean_per_dispatch_kb,total_kb\n")
                for n in sorted(agg, key=lambda n: -sum(agg[n])):
                    f.write("%s,%s,%d,%.1f,%.0f\n" % (n, ctr, len(agg[n]), sum(agg[n]) / len(agg[n]), sum(agg[n])))
            v = [float(r["Counter_Value"]) for r in rows_full if short(r["Kernel_Name"]) == name and r["Grid_Size"] == grid]
            v = split_upper(v) if ctr == "FETCH_SIZE" else v
            rec[ctr.lower() + "_kb_raw"] = {"dispatches": len(v), "mean": statistics.mean(v) if v else None, "values": [round(x, 1) for x in v[:400]]}
        out[c] = rec
        json.dump(rec, open(os.path.join(S, "dominant_launch_cfg%d.json" % c), "w"), indent=1)
        # the time-dominant kernel FAMILY: every dispatch of it in the kernel trace (durations) and in the counter passes
        fam = {"kernel": family, "variants": {}}
        rows = [r for r in csv.DictReader(open(kt)) if short(r["Kernel_Name"]).startswith(family)]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        tot_all = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)))
        if dur:
            fam.update(dispatches=len(dur), mean_us=statistics.mean(dur), median_us=statistics.median(dur), total_us=sum(dur),
                       share_of_kernel_time=sum(dur) / tot_all if tot_all else None)
            by = collections.defaultdict(list)
            for r in rows:
                by[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            fam["variants"] = {k: {"dispatches": len(v), "mean_us": statistics.mean(v)} for k, v in by.items()}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cc = one(os.path.join(O, "pmc_cfg%d_%s" % (c, ctr), "**", "*counter_collection.csv"))
            if cc:
                v = [float(r["Counter_Value"]) for r in csv.DictReader(open(cc)) if short(r["Kernel_Name"]).startswith(family)]
                if v:
                    fam[ctr.lower() + "_kb_raw"] = {"dispatches": len(v), "mean": statistics.mean(v)}
        json.dump(fam, open(os.path.join(S, "dominant_kernel_cfg%d.json" % c), "w"), indent=1)
    mf = one(os.path.join(O, "pmc_cfg2_mfma", "**", "*counter_collection.csv"))
    if mf:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(mf)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(S, "bench_cfg2_pmc_mfma.csv"), "w") as f:
            f.write("kernel,counter,dispatches,mean_per_dispatch,total\n")
            for n in sorted(agg, key=lambda n: -sum(sum(v) for v in agg[n].values())):
                for cn, v in sorted(agg[n].items()):
                    f.write("%s,%s,%d,%.1f,%.0f\n" % (n, cn, len(v), sum(v) / len(v), sum(v)))
    print("summarised", {c: (r["kernel"], r["dispatches"], round(r["median_us"], 2)) for c, r in out.items()})


def install(O):
    S, P = os.path.join(O, "summary"), os.path.join(ROOT, "profiles")
    bid = build_id()
    cp = [("bench_cfg2_result.json", TAG + "_bench_cfg2_result.json"), ("bench_cfg3_result.json", TAG + "_bench_cfg3_result.json"),
          ("bench_cfg4_result.json", TAG + "_bench_cfg4_result.json"), ("bench_cfg5_result.json", TAG + "_bench_cfg5_result.json"),
          ("bench_cfg2_zipf_result.json", TAG + "_bench_cfg2_zipf_ids_result.json"), ("step_table_cfg2.txt", TAG + "_step_table_cfg2.txt"),
          ("host_time_supernet.txt", TAG + "_host_time_supernet.txt"), ("supernet_fullpath_step.txt", TAG + "_supernet_fullpath_step.txt"),
          ("parser_bench.txt", TAG + "_parser_bench.txt"), ("e2e_tsv_run.txt", TAG + "_e2e_tsv_run.txt"), ("dedup_cost.txt", TAG + "_dedup_cost_global_batch.txt"),
          ("bench_cfg2_dp_path_result.json", TAG + "_bench_cfg2_dp_path_result.json"), ("bench_cfg2_dp_path_eager_result.json", TAG + "_bench_cfg2_dp_path_eager_result.json"),
          ("dp_overhead.txt", TAG + "_dp_overhead.txt"), ("bench_cfg2_row_sharded_result.json", TAG + "_bench_cfg2_row_sharded_result.json"),
          ("ab_level_balance.txt", TAG + "_ab_level_balance.txt"), ("supernet_step_table_cfg3.txt", TAG + "_supernet_step_table_cfg3.txt"),
          ("gemm_fast_stamps.txt", TAG + "_gemm_fast_stamps.txt"), ("gemm_vs_vendor.txt", TAG + "_gemm_vs_vendor.txt"),
          ("bench_cfg2_driver_flags_result.json", TAG + "_bench_cfg2_driver_flags_result.json"),
          ("bench_cfg2_dp_path_real_result.json", TAG + "_bench_cfg2_dp_path_real_collectives_result.json"), ("ab_fuse_final.txt", TAG + "_ab_fuse_final.txt"),
          ("gpu_tests.txt", TAG + "_gpu_tests.txt"),
          ("ab_graph_vs_launch.txt", TAG + "_ab_graph_vs_launch.txt"), ("launch_gaps.txt", TAG + "_launch_gaps.txt"),
          ("search_operating_point.txt", TAG + "_search_operating_point_final.txt"), ("ab_persist_uc.txt", TAG + "_ab_persist_uc_final.txt"),
          ("persist_timeline.txt", TAG + "_persist_timeline_final.txt"), ("seam_probe.txt", TAG + "_seam_probe_final.txt"),
          ("anyorder_probe.txt", TAG + "_anyorder_probe.txt")]
    for a, b in cp:
        src = os.path.join(O, a)
        if os.path.exists(src) and os.path.getsize(src) > 0:
            txt = open(src, errors="replace").read()
            txt = "\n".join(l for l in txt.splitlines() if "amdgpu.ids" not in l and not l.startswith(("ROCm version", "Hostname", "Librccl", "HIP version", "RCCL version"))) + "\n"
            open(os.path.join(P, b), "w").write(txt)
    for c in CFGS:
        for a, b in (("bench_cfg%d_kernel_stats.csv" % c, TAG + "_bench_cfg%d_kernel_stats.csv" % c), ("bench_cfg%d_pmc_fetch.csv" % c, TAG + "_bench_cfg%d_pmc_fetch.csv" % c),
                     ("bench_cfg%d_pmc_write.csv" % c, TAG + "_bench_cfg%d_pmc_write.csv" % c), ("dominant_launch_cfg%d.json" % c, TAG + "_dominant_launch_cfg%d_dispatches.json" % c)):
            if os.path.exists(os.path.join(S, a)):
                shutil.copy(os.path.join(S, a), os.path.join(P, b))
        up = os.path.join(O, "bench_cfg%d_under_profiler.out" % c)
        if os.path.exists(up):
            lines = [l for l in open(up, errors="replace").read().splitlines() if l.startswith("{")]
            if lines:
                open(os.path.join(P, TAG + "_bench_cfg%d_under_profiler.json" % c), "w").write(lines[-1] + "\n")
        dl, res = os.path.join(S, "dominant_launch_cfg%d.json" % c), os.path.join(O, "bench_cfg%d_result.json" % c)
        if os.path.exists(dl) and os.path.exists(res):
            d, r = json.load(open(dl)), json.loads(open(res).read().strip().splitlines()[-1])
            f, w = d.get("fetch_size_kb_raw", {}).get("mean"), d.get("write_size_kb_raw", {}).get("mean")
            if f is not None and w is not None:
                big = r.get("roofline_largest_gemm", r["roofline"])
                j = {"build_id": bid, "kernel": big["kernel"], "kernels_of_the_launch": "%s, grid size %s threads" % (d["kernel"], d["grid_size"]),
                     "source": "tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `python3 bench.py%s --steps 12 --warmup 3 --steps-only%s`; per-dispatch rows of this launch in profiles/%s_dominant_launch_cfg%d_dispatches.json" % (
                         "" if c == 2 else " --config %d" % c, " --no-graph" if c == 2 else "", TAG, c),
                     "fetch_size_kb_raw": f, "fetch_correction": 2.0,
                     "fetch_correction_source": "MI355X_MICROARCH.md HBM section (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); confirmed on known byte counts in round 1 (profiles/r01_pmc_fetch_calibration_loadrate.csv)",
                     "write_size_kb": w, "traffic_bytes_per_launch": int(round((2 * f + w) * 1024)),
                     "algorithmic_bytes_per_launch": big.get("algorithmic_bytes"),
                     "per_dispatch_duration_us": {"median": d["median_us"], "mean": d["mean_us"], "dispatches": d["dispatches"], "source": "rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline`"}}
                if c == 2:
                    j["note"] = ("FETCH_SIZE counts what the 8 private L2s request from the fabric (Infinity Cache included), not HBM reads: every launch starts on cold "
                                 "L2s, and each XCD needs the 128 rows of x and 192 rows of W of its 4 x 6 block of output tiles = 2.0 MB for K = 1565, "
                                 "the minimum for 8 equal blocks -> 8 x 2.0 = 15.8 MB of fabric reads against 6.4 MB of operands; HBM sees each operand byte at most once")
                json.dump(j, open(os.path.join(P, "dominant_gemm_traffic_cfg%d.json" % c), "w"), indent=1)
    for c in CFGS:
        fk, res = os.path.join(S, "dominant_kernel_cfg%d.json" % c), os.path.join(O, "bench_cfg%d_result.json" % c)
        if os.path.exists(fk) and os.path.exists(res):
            d, r = json.load(open(fk)), json.loads(open(res).read().strip().splitlines()[-1])
            shutil.copy(fk, os.path.join(P, TAG + "_dominant_kernel_cfg%d_dispatches.json" % c))
            f, w = d.get("fetch_size_kb_raw", {}).get("mean"), d.get("write_size_kb_raw", {}).get("mean")
            if f is not None and w is not None:
                json.dump({"build_id": bid, "kernel": r["roofline"]["kernel"], "fetch_size_kb_raw_mean_per_dispatch": f, "fetch_correction": 2.0,
                           "write_size_kb_mean_per_dispatch": w, "traffic_bytes_per_launch": int(round((2 * f + w) * 1024)),
                           "algorithmic_bytes_per_launch": r["roofline"].get("algorithmic_bytes"),
                           "per_dispatch_duration_us": {"mean": d.get("mean_us"), "median": d.get("median_us"), "dispatches": d.get("dispatches"),
                                                        "share_of_kernel_time": d.get("share_of_kernel_time")},
                           "source": "tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), mean over every "
                                     "dispatch of the kernel (all instantiations); durations from the rocprofv3 --kernel-trace run of the default bench",
                           "note": "fabric traffic of the 8 private L2s (Infinity Cache included), not HBM reads: every launch starts on cold L2s"},
                          open(os.path.join(P, "dominant_kernel_traffic_cfg%d.json" % c), "w"), indent=1)
    for ps in ("A", "B"):
        f = os.path.join(O, "mha_pmc", "mha_pmc_pass%s.csv" % ps)
        if os.path.exists(f) and os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(P, TAG + "_mha_pmc_pass%s.csv" % ps))
    if os.path.exists(os.path.join(S, "bench_cfg2_pmc_mfma.csv")):
        shutil.copy(os.path.join(S, "bench_cfg2_pmc_mfma.csv"), os.path.join(P, TAG + "_bench_cfg2_pmc_mfma.csv"))
    print("installed into profiles/ for build", bid)


if __name__ == "__main__":
    (summarise if sys.argv[1] == "--summarise" else install)(sys.argv[2])
