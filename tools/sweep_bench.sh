# A/B of scheduler knobs on one box: bash tools/sweep_bench.sh  (each line: environment => registrations/s, ms per step)
run() { echo -n "$1 $2 => "; env $1 timeout 300 python bench.py --no-cpu --dense-pairs 0 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"; }
run "X=1" ""
run "NDTGPU_MATCH_GROUPS=240" ""
run "NDTGPU_MATCH_GROUPS=224" ""
run "NDTGPU_MATCH_GROUPS=208" ""
run "NDTGPU_MATCH_GROUPS=192" ""
run "NDTGPU_MATCH_GROUPS=224 NDTGPU_SLOTS=3" ""
run "X=1" ""
