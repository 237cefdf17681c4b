cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 300 python bench.py --config 5 2>&1 | tail -3 | cut -c1-1800
timeout 300 python bench.py --config fuse 2>&1 | tail -3 | cut -c1-2200
