#!/usr/bin/env python3
"""Turns gpurun_out/prof_* (rocprofv3 CSV) into the committed profiles/rNN_* summaries.
usage: python tools/summarize_profiles.py r02"""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
# (on the GPU box the summaries go under gpurun_out/ -- the only directory that travels back -- and are copied to profiles/ here)
pr = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
# kernel stats
rows = list(csv.DictReader(open(os.path.join(go, "prof_kt", "bench_kernel_stats.csv"))))
with open(os.path.join(pr, "%s_bench_kernel_stats.csv" % tag), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(list(rows[0].keys()))
    for r in rows[:20]:
        w.writerow([(v[:110] if isinstance(v, str) else v) for v in r.values()])
# (the stream-fed matcher's instances, its publish / wait kernels and the pack kernels are not "a launch per step": listed in
#  the CSV, not in the per-launch summary)
ours = {r["Name"][:24]: r for r in rows if ("ndt_build" in r["Name"] or "ndt_match_kernel" in r["Name"])}
stream_rows = [r for r in rows if "ndt_match_stream" in r["Name"] or "ndt_stream_" in r["Name"]]
serial = {}
sp = os.path.join(go, "prof_kt_serial", "bench_kernel_stats.csv")
if os.path.exists(sp):
    srows = list(csv.DictReader(open(sp)))
    with open(os.path.join(pr, "%s_bench_kernel_stats_serial.csv" % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(list(srows[0].keys()))
        for r in srows[:20]:
            w.writerow([(v[:110] if isinstance(v, str) else v) for v in r.values()])
    serial = {("ndt_build_kernel" if "build" in r["Name"] else "ndt_match_kernel"): r for r in srows if "ndt_build" in r["Name"] or "ndt_match" in r["Name"]}
# PMC
def pmc(dirname):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(go, dirname, "bench_counter_collection.csv"))):
        k = "ndt_build_kernel" if "ndt_build" in r["Kernel_Name"] else "ndt_match_kernel" if "ndt_match_kernel" in r["Kernel_Name"] else r["Kernel_Name"][:30]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
fetch, write, sq = pmc("prof_fetch"), pmc("prof_write"), pmc("prof_sq")
sys.path.insert(0, root)
from ndt_feature_graph_amd import binding
out = {"round": tag, "lib_version": binding.lib().ndtgpu_version().decode(),   # bench.py refuses a summary of another binary
       "command": "python bench.py --steps 10 --warmup 2 --no-cpu --dense-pairs 0 --no-pipeline for the PMC passes and *_serial (1024 pairs x 100k pts per step: one build launch of 2048 scans + one matcher launch of 1024 pairs, alone on the chip); avg_ns / calls: the same command without --no-pipeline (the registrar's default form: builds beside ONE running instance of ndt_match_stream_kernel; its ndt_match_kernel launches are the bench's serial measuring steps)",
       "note": "FETCH_SIZE / WRITE_SIZE are in KB per dispatch (rocprofv3, separate --pmc passes, --kernel-include-regex ndt_). "
               "On gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM): "
               "hbm_read_bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is taken at face value (uncalibrated).",
       "kernels": {}}
for k in ("ndt_build_kernel", "ndt_match_kernel"):
    f_kb, w_kb = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
    out["kernels"][k] = {"FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "hbm_read_bytes": 2 * f_kb * 1024,
                         "hbm_write_bytes": w_kb * 1024, "hbm_bytes_per_launch": 2 * f_kb * 1024 + w_kb * 1024,
                         "sq_per_dispatch": sq.get(k, {})}
for name, r in ours.items():
    key = "ndt_build_kernel" if "build" in r["Name"] else "ndt_match_kernel"
    out["kernels"][key]["avg_ns"] = float(r["AverageNs"]); out["kernels"][key]["calls"] = int(r["Calls"])
for key, r in serial.items():
    out["kernels"][key]["avg_ns_serial"] = float(r["AverageNs"])
out["stream_fed"] = [{"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"])} for r in stream_rows]
json.dump(out, open(os.path.join(pr, "%s_pmc_traffic.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])


# ---- the 3D (--config 5) and node-map (--config fuse) benches: kernel statistics + per-kernel counters -------------
import re
def short(name):
    # the build kernel keeps its template arguments: <stride, MODE, nice, SCAT> are different kernels (accumulate /
    # moments -> Gaussians / ranking), and the 3D build is judged launch by launch
    n = name.replace("void ", "").split("(")[0].strip()
    return n if n.startswith("ndt_build_kernel") else re.sub(r"<.*", "", n)
for c in ("5", "fuse"):
    kt = os.path.join(go, "prof_kt_%s" % c, "bench_kernel_stats.csv")
    if not os.path.exists(kt):
        continue
    rows = list(csv.DictReader(open(kt)))
    with open(os.path.join(pr, "%s_c%s_kernel_stats.csv" % (tag, c)), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(list(rows[0].keys()))
        for r in rows[:24]:
            w.writerow([(v[:110] if isinstance(v, str) else v) for v in r.values()])
    def pmc_all(dirname):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        fn = os.path.join(go, dirname, "bench_counter_collection.csv")
        if not os.path.exists(fn):
            return {}
        for r in csv.DictReader(open(fn)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return {k: {cn: sum(v) / len(v) for cn, v in d.items()} for k, d in agg.items()}
    fe, wr, sq_ = pmc_all("prof_fetch_%s" % c), pmc_all("prof_write_%s" % c), pmc_all("prof_sq_%s" % c)
    res = {"round": tag, "lib_version": out["lib_version"], "command": "python bench.py --config %s" % c,
           "note": out["note"] + "  Averages over ALL dispatches of a kernel name in the run (several launch shapes of ndt_build_kernel share a name).",
           "kernels": {}}
    for k in sorted(set(fe) | set(wr) | set(sq_)):
        d = {"sq_per_dispatch": sq_.get(k, {})}
        if k in fe and k in wr:
            d.update({"FETCH_SIZE_KB": fe[k]["FETCH_SIZE"], "WRITE_SIZE_KB": wr[k]["WRITE_SIZE"],
                      "hbm_bytes_per_launch": 2 * fe[k]["FETCH_SIZE"] * 1024 + wr[k]["WRITE_SIZE"] * 1024})
        q = d["sq_per_dispatch"]
        if q.get("SQ_WAVE_CYCLES"):
            d["wait_any_over_wave_cycles"] = q.get("SQ_WAIT_ANY", 0) / q["SQ_WAVE_CYCLES"]
            d["valu_per_wave"] = q.get("SQ_INSTS_VALU", 0) / max(1.0, q.get("SQ_WAVES", 1))
        for r in rows:
            if short(r["Name"]) == k:
                d.setdefault("avg_ns_by_shape", []).append([r["Name"][:70], float(r["AverageNs"]), int(r["Calls"])])
        res["kernels"][k] = d
    json.dump(res, open(os.path.join(pr, "%s_c%s_pmc.json" % (tag, c)), "w"), indent=1)
    print("config", c, "->", list(res["kernels"].keys()))
