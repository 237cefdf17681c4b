# A/B of library variants / scheduler knobs on one box: bash tools/sweep_bench.sh  (each line: setting => registrations/s, ms per step)
run() { echo -n "$1 $2 => "; env $1 timeout 300 python bench.py --no-cpu --dense-pairs 0 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['ms_isolated'],3) for k,v in d['kernels'].items()})"; }
V=$PWD/ndt_feature_graph_amd/variants
run "X=1" ""
for v in u3 u4 u5 u4d; do run "NDTGPU_LIB=$V/libndtgpu_$v.so" ""; done
run "X=1" ""
