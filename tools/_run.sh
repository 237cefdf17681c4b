cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export PYTHONUNBUFFERED=1
./tests/native/host_demo 2>&1 | tail -25
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
