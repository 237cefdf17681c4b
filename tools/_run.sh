cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "odometry_cells" 2>&1 | tail -15
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
