#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, then the matcher / build micro-benchmarks for the shipped library and the
# variants under ndt_feature_graph_amd/variants/ (A/B on the SAME box), the matcher timeline, then the bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ulimit -c 0
export PYTHONUNBUFFERED=1
tag=${1:-r03a}
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
timeout 300 python tools/microbench.py > gpurun_out/${tag}_mb_new.log 2>&1
NDTGPU_SLOTS=1 timeout 300 python tools/microbench.py > gpurun_out/${tag}_mb_new_s1.log 2>&1
NDTGPU_DOUBLE_THRESH=0 timeout 300 python tools/microbench.py > gpurun_out/${tag}_mb_new_dbl0.log 2>&1
for v in ndt_feature_graph_amd/variants/*.so; do
  n=$(basename $v .so)
  case $n in *_tl) NDTGPU_LIB=$PWD/$v timeout 300 python tools/timeline_match.py > gpurun_out/${tag}_timeline.log 2>&1;;
             *) NDTGPU_LIB=$PWD/$v timeout 300 python tools/microbench.py > gpurun_out/${tag}_mb_${n}.log 2>&1;; esac
done
timeout 600 python bench.py --no-cpu > gpurun_out/${tag}_bench.log 2>&1
NDTGPU_DOUBLE_THRESH=0 timeout 600 python bench.py --no-cpu > gpurun_out/${tag}_bench_dbl0.log 2>&1
timeout 300 python bench.py --no-cpu --no-pipeline > gpurun_out/${tag}_bench_serial.log 2>&1
grep -H "^match full\|cycles:\|pair total" gpurun_out/${tag}_mb_*.log | sed 's/gpurun_out\///'
grep -v amdgpu.ids gpurun_out/${tag}_timeline.log
for f in bench bench_dbl0 bench_serial; do echo $f; tail -1 gpurun_out/${tag}_$f.log | cut -c1-330; done
