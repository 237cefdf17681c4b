#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + HBM PMC counters of the bench command, and of the
# 3D (--config 5) and node-map (--config fuse) benches.
# PMC passes are separate (--pmc never combined with trace domains) and restricted to our kernels.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ulimit -c 0
CMD="python bench.py --steps 10 --warmup 2 --no-cpu --dense-pairs 0"
$CMD > gpurun_out/bench_warm.log 2>&1      # (a fresh box pages the image in for a minute: not inside a trace)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o bench -- $CMD > gpurun_out/bench_kt.log 2>&1
# the same workload without the pipeline: kernel durations with the chip to themselves
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt_serial -o bench -- $CMD --no-pipeline > gpurun_out/bench_kt_serial.log 2>&1
# PMC passes on the SERIAL form: every dispatch is then one build launch of 2048 scans / one matcher launch of 1024 pairs (in the
# default form one instance of the stream-fed matcher serves many batches: its counters are not "per launch" of anything)
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "ndt_" --output-format csv -d gpurun_out/prof_fetch -o bench -- $CMD --no-pipeline > gpurun_out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "ndt_" --output-format csv -d gpurun_out/prof_write -o bench -- $CMD --no-pipeline > gpurun_out/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --kernel-include-regex "ndt_" --output-format csv -d gpurun_out/prof_sq -o bench -- $CMD --no-pipeline > gpurun_out/bench_sq.log 2>&1
for c in 5 fuse; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt_$c -o bench -- python bench.py --config $c > gpurun_out/bench_kt_$c.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "ndt_" --output-format csv -d gpurun_out/prof_fetch_$c -o bench -- python bench.py --config $c > gpurun_out/bench_fetch_$c.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "ndt_" --output-format csv -d gpurun_out/prof_write_$c -o bench -- python bench.py --config $c > gpurun_out/bench_write_$c.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --kernel-include-regex "ndt_" --output-format csv -d gpurun_out/prof_sq_$c -o bench -- python bench.py --config $c > gpurun_out/bench_sq_$c.log 2>&1
  python bench.py --config $c > gpurun_out/bench_c$c.log 2>&1
done
python bench.py > gpurun_out/bench_full.log 2>&1
# summarise here (the raw traces and per-dispatch counter tables exceed what travels back), keep the small tables only
python tools/summarize_profiles.py r06 gpurun_out/r06_profiles > gpurun_out/r06_summarize.log 2>&1
python tools/kernel_resources.py > gpurun_out/r06_profiles/r06_kernel_resources.txt 2>&1
for d in prof_kt prof_kt_serial prof_kt_5 prof_kt_fuse; do mkdir -p gpurun_out/r06_raw/$d; cp gpurun_out/$d/bench_kernel_stats.csv gpurun_out/r06_raw/$d/ 2>/dev/null; done
rm -rf gpurun_out/prof_*
tail -1 gpurun_out/bench_full.log | cut -c1-400
tail -1 gpurun_out/bench_c5.log | cut -c1-400
tail -1 gpurun_out/bench_cfuse.log | cut -c1-400
ls gpurun_out/r06_profiles; tail -3 gpurun_out/r06_summarize.log | cut -c1-300
