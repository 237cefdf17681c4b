cd "${GRAFT_REPO_ROOT:-/root/repo}"; ulimit -c 0; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
NDTGPU_LIB=$PWD/ndt_feature_graph_amd/variants/libndtgpu_x_new.so timeout 300 python tools/timeline_match.py --pairs 256 2>&1 | grep -v amdgpu.ids
