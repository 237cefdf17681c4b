cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export PYTHONUNBUFFERED=1
for p in 0 2 4 6 9; do for d in default 0; do
  if [ $d = default ]; then unset NDTGPU_DOUBLE_THRESH; else export NDTGPU_DOUBLE_THRESH=$d; fi
  echo "park $p dbl $d: $(NDTGPU_PARK_ITERS=$p timeout 120 python tools/repro_match.py 1024 4 2>&1 | grep 'match [123] ok' | awk '{print $4}' | tr '\n' ' ')"
done; done
unset NDTGPU_DOUBLE_THRESH
echo "slots1: $(NDTGPU_SLOTS=1 timeout 120 python tools/repro_match.py 1024 4 2>&1 | grep 'match [123] ok' | awk '{print $4}' | tr '\n' ' ')"
