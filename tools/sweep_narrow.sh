run() { echo -n "$1 $2 => "; env $1 timeout 300 python bench.py --no-cpu --dense-pairs 0 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['ms_isolated'],3) for k,v in d['kernels'].items()})"; }
V=$PWD/ndt_feature_graph_amd/variants
run "X=1" ""
run "NDTGPU_LIB=$V/libndtgpu_m256.so NDTGPU_SLOTS=1 NDTGPU_MATCH_GROUPS=512" ""
run "NDTGPU_LIB=$V/libndtgpu_m256.so NDTGPU_SLOTS=1 NDTGPU_MATCH_GROUPS=512 NDTGPU_DOUBLE_THRESH=0" ""
run "NDTGPU_LIB=$V/libndtgpu_m256.so NDTGPU_SLOTS=2 NDTGPU_MATCH_GROUPS=256" ""
run "NDTGPU_LIB=$V/libndtgpu_m256.so NDTGPU_SLOTS=1 NDTGPU_MATCH_GROUPS=512" "--buffers 4"
run "X=1" ""
NDTGPU_LIB=$V/libndtgpu_m256.so NDTGPU_SLOTS=1 NDTGPU_MATCH_GROUPS=512 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
