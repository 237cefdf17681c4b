export NDTGPU_REG_VERBOSE=1
for v in default norecal; do
  extra=""; [ $v = norecal ] && extra="--registrar recalibrate_pct=-1"
  for rep in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu --dense-pairs 0 $extra > gpurun_out/r6B_$v.json 2> gpurun_out/r6B_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6B_$v.json').read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['ms_per_step'],3), d['config'].get('registrar'), {k:(round(v['ms_per_launch'],3), round(v['ms_isolated'],3)) for k,v in d['kernels'].items()})
PY
  done
done
python -m pytest tests/test_gpu_registrar_fullsize.py -x -q -k "clutter or priority" 2>&1 | tail -5
