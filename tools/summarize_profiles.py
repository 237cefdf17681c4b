#!/usr/bin/env python3
"""Turns gpurun_out/prof_* (rocprofv3 CSV) into the committed profiles/rNN_* summaries.
usage: python tools/summarize_profiles.py r02"""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
# kernel stats
rows = list(csv.DictReader(open(os.path.join(go, "prof_kt", "bench_kernel_stats.csv"))))
with open(os.path.join(pr, "%s_bench_kernel_stats.csv" % tag), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(list(rows[0].keys()))
    for r in rows[:20]:
        w.writerow([(v[:110] if isinstance(v, str) else v) for v in r.values()])
ours = {r["Name"][:24]: r for r in rows if "ndt_" in r["Name"]}
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
        k = "ndt_build_kernel" if "ndt_build" in r["Kernel_Name"] else "ndt_match_kernel" if "ndt_match" in r["Kernel_Name"] else r["Kernel_Name"][:30]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
fetch, write, sq = pmc("prof_fetch"), pmc("prof_write"), pmc("prof_sq")
sys.path.insert(0, root)
from ndt_feature_graph_amd import binding
out = {"round": tag, "lib_version": binding.lib().ndtgpu_version().decode(),   # bench.py refuses a summary of another binary
       "command": "python bench.py --steps 10 --warmup 2 --no-cpu (1024 pairs x 100k pts per step: one build launch of 2048 scans + one matcher launch, three-buffer pipeline; *_serial: same with --no-pipeline)",
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
json.dump(out, open(os.path.join(pr, "%s_pmc_traffic.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
