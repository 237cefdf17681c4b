cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/microbench.py 2>&1 | grep "^build\|^match full"
timeout 120 python tools/latency_probe.py 2>&1 | grep "single 2D\|back to back"
timeout 300 python bench.py --config 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 build', d['kernels']['build_64_sweeps'], 'pair', d['single_pair'])"
timeout 300 python bench.py --no-cpu 2>&1 | tail -1 | cut -c1-300
