cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export PYTHONUNBUFFERED=1
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/r03w_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03w_bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['frac_isolated'])
for k,v in d['single_pair_latency'].items(): print(k[:40], v['gpu_ms'], v['gpu_ms_min_max'], v['speedup'])
print(d['pcie_inclusive']['value'])
PY
timeout 120 python tools/latency_probe.py 2>&1 | grep "single 2D\|back to back"
