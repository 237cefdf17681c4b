"""The RCCL branch, executed (-m gpu): NDTGPU_FORCE_COLLECTIVES=1 makes a ONE-rank run create the `nccl` process group and go
through every collective of the path -- the two all_gather_into_tensor calls of the edge results after every step of
bench.py, `exchange_node_maps` (the all-gather of the packed node maps, SURVEY.md 8e phase B) and `gather_edge_results`
(phase D) of bench.py --config 4 -- instead of the world == 1 short cuts.  What an 8-GPU node would run first, run here on
one GPU: API use, stream order against the registrar's internal streams, buffer layouts, result order.

Each case is its own process (a process group cannot be re-created cleanly inside the pytest process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, port, timeout=900):
    env = dict(os.environ, NDTGPU_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return p.stdout


def bench_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert lines, "no bench line in: " + out[-3000:]
    return json.loads(lines[-1])


def test_bench_headline_path_through_rccl_on_one_rank():
    out = run(["bench.py", "--steps", "4", "--warmup", "1", "--pairs", "96", "--points", "20000", "--no-cpu", "--dense-pairs", "0"], 29611)
    line = bench_line(out)
    c = line["collectives"]
    assert c["process_group"] == "nccl (RCCL)" and c["forced_on_one_rank"] and c["all_gather_into_tensor_calls"] >= 8
    assert c["gathered_rows_equal_local"] is True
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["kernels"]["ndt_match_kernel"]["converged_frac"] > 0.8


def test_bench_config4_exchange_and_gather_through_rccl_on_one_rank():
    out = run(["bench.py", "--config", "4", "--nodes", "150", "--scans-per-node", "2", "--node-points", "5000", "--no-all-pairs",
               "--steps", "2", "--warmup", "1"], 29612)
    line = bench_line(out)
    c = line["collectives"]
    assert c["process_group"] == "nccl (RCCL)" and c["forced_on_one_rank"] and c["gathered_rows_equal_local"] is True
    assert line["value"] > 0


SCRIPT = r"""
import numpy as np, torch, torch.distributed as dist
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import distributed as D, synth
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert D.collectives_forced()
# phase A-B: 12 node maps built, packed, exchanged through all_gather_into_tensor, unpacked into a second set
pr = synth.pair_2d(torch.arange(1, 13, dtype=torch.int64, device=dev), 20000, device=dev)
a = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=12, max_cells=4096); a.enable_occupancy()
b = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=12, max_cells=4096); b.enable_occupancy()
st = torch.cuda.current_stream()
a.build(pr["fixed"].contiguous(), range_limit=30.0, stream=st)
cap = 1024
buf = torch.zeros((12, a.pack_bytes(cap, True)), dtype=torch.uint8, device=dev)
a.pack_cells(buf, 0, 12, cells_cap=cap, with_occupancy=True, stream=st)
rec = D.exchange_node_maps(buf, 12, 0, 1)
assert rec.data_ptr() != buf.data_ptr() and torch.equal(rec, buf)          # it went through the collective, unchanged
b.unpack_cells(rec, 0, 12, with_occupancy=True, stream=st)
torch.cuda.synchronize()
for k in (0, 5, 11):
    for x, y in zip(a.export_cells(k), b.export_cells(k)):
        assert np.array_equal(x, y)
# phase D: edge results of a block-cyclic shard back in edge order
n = 1000
T = torch.arange(n * 16, dtype=torch.float64, device=dev).reshape(n, 16)
R = (torch.arange(n * 64, device=dev) % 251).to(torch.uint8).reshape(n, 64)
mine, Tg, Rg = D.register_sharded(n, 0, 1, lambda e: (T[torch.as_tensor(e, device=dev)], R[torch.as_tensor(e, device=dev)]), 256)
assert Tg.data_ptr() != T.data_ptr() and torch.equal(Tg, T) and torch.equal(Rg, R) and len(mine) == n
dist.barrier(); dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def test_exchange_and_gather_functions_through_rccl_on_one_rank():
    out = run(["-c", SCRIPT], 29613)
    assert "RCCL_ONE_RANK_OK" in out
