cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export PYTHONUNBUFFERED=1
timeout 600 python bench.py --config 4 --nodes 1000 --gated 2>&1 | tail -2 | cut -c1-3000
timeout 1200 python bench.py --config 4 2>&1 | tail -2 | cut -c1-3000
