"""BASELINE configs[2] through the entry bench.py TIMES: ndtgpu_register_batch_device on a default registrar (-m gpu).

The headline number is measured on a registrar with eight internal map sets whose matcher is the stream-fed form (ONE running
instance on a measured share of the CUs, grid builds on two streams beside it).  This file puts exactly
that path under the oracle at full size -- 1024 pairs x 100 k points per call, >= 24 calls without a host wait, output
buffers in rotation behind tickets:
  * every call's poses and deterministic result fields are the BITS of ndtgpu_mapset_build + ndtgpu_match_batch_device
    (the two-call path that tests/test_gpu_fullsize.py checks against the oracle on its own),
  * 32 sampled pairs, the 8 longest registrations among them, against the CPU oracle (1e-4 m / 1e-4 rad, BASELINE.json),
  * a registrar whose split of the chip is FORCED away from the measured one gives the same bits,
  * a registrar fed alternating batches of the plain halls and of the cluttered "dense" scene (five times the cells per
    map) gives, batch by batch, the bits of the two-call path, and re-measures its split instead of keeping a stale one.
Reference call shape: NDTFeatureGraph::updateLinksUsingNDTRegistration, ndt_feature/src/ndt_feature_src/ndt_feature_graph.cpp:347-353."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
DET_FIELDS = ["converged", "iterations", "fevals", "exit_code", "score", "n_source", "n_target", "pair_terms_g", "pair_terms_h"]
RES, SIZE, RNG = 0.5, [100.0, 100.0, 1.0], 30.0
B, NP = 1024, 100000


@pytest.fixture(scope="module")
def N():
    import ndt_feature_graph_amd as N
    if N.device_count() < 1:
        pytest.fail("no HIP device visible: the HIP path cannot run (there is no CPU fallback)")
    return N


@pytest.fixture(scope="module")
def O():
    import oracle
    return oracle


def two_call(N, both, T0, n):
    """ndtgpu_mapset_build (ONE launch for the 2 n scans, as the registrar's sub-batch) + ndtgpu_match_batch_device"""
    import torch
    from ndt_feature_graph_amd import binding
    dev = both.device
    ms = N.MapSet(RES, [0, 0, 0], SIZE, n_maps=2 * n, max_cells=4096)
    st = torch.cuda.current_stream()
    ms.build(both, range_limit=RNG, stream=st)
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    T16 = T0.clone()
    res = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    binding.match_batch_device(ms, idx, ms, idx + n, T16, res, n, stream=st)
    torch.cuda.synchronize()
    out = T16.cpu().numpy(), res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(n)
    ms.close()
    return out


@pytest.fixture(scope="module")
def halls(N):
    """the headline batch in HBM (fixed scans, then moving scans, in ONE tensor like bench.py) + its two-call result"""
    import torch
    from ndt_feature_graph_amd import synth
    dev = torch.device("cuda", 0)
    pr = synth.pair_2d(torch.arange(1, 1 + B, dtype=torch.int64, device=dev), NP, device=dev, chunk_bytes=2 << 30)
    both = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
    pr["fixed"] = pr["moving"] = None
    T0 = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
    T_ref, r_ref = two_call(N, both, T0, B)
    return {"both": both, "T0": T0, "T_init": pr["T_init"].cpu().numpy(), "T_ref": T_ref, "r_ref": r_ref, "dev": dev}


def same_bits(binding, T16, res, T_ref, r_ref, what=""):
    assert np.array_equal(T16.cpu().numpy(), T_ref), "poses " + what
    r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(-1)
    for f in DET_FIELDS:
        assert np.array_equal(r[f], r_ref[f]), f + " " + what
    return r


def run_calls(reg, torch, both, T0, n_calls, n_out):
    """n_calls submits back to back on one caller stream, n_out output buffers in rotation: a buffer is read (cloned, on the
    stream that waited for its ticket) right before the call that overwrites it.  No host wait until the end."""
    dev = both.device
    n = T0.shape[0]
    st = torch.cuda.Stream(device=dev)
    outs = [(T0.clone(), torch.zeros((n, 64), dtype=torch.uint8, device=dev)) for _ in range(n_out)]
    tickets = [0] * n_out
    kept = []
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for k in range(n_calls):
            T16, res = outs[k % n_out]
            if tickets[k % n_out]:
                reg.wait_stream(st, ticket=tickets[k % n_out])
                kept.append((T16.clone(), res.clone()))
            T16.copy_(T0)
            tickets[k % n_out] = reg.submit(both[:n], both[n:], T16, res, range_limit=RNG, stream=st)
    reg.sync()
    # the buffers still in flight at the end, oldest first
    order = sorted(range(n_out), key=lambda i: tickets[i])
    kept += [(outs[i][0].clone(), outs[i][1].clone()) for i in order if tickets[i]]
    torch.cuda.synchronize()
    assert len(kept) == n_calls
    return kept


def test_config3_through_the_registrar_full_size(N, O, halls):
    """26 calls of 1024 x 100 k pairs through a DEFAULT registrar (depth 8, stream-fed matcher, measured split, two build
    streams), 8 output buffers behind tickets: every call = the bits of build + match_batch_device; 32 oracle samples."""
    import torch
    from ndt_feature_graph_amd import binding
    both, T0, T_ref, r_ref = halls["both"], halls["T0"], halls["T_ref"], halls["r_ref"]
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=B, max_cells=4096)            # depth: the default (8)
    assert reg.depth == 8
    kept = run_calls(reg, torch, both, T0, n_calls=26, n_out=8)
    info = reg.info()
    assert info["matcher_form"] == binding.MATCHER_STREAM_FED and info["build_streams"] == 2
    assert info["calibrations"] == 1 and 64 <= info["matcher_groups"] <= 224, info         # measured once, a plausible share
    assert info["submitted"] == 26 and 100 < info["cells_per_map"] < 2000
    assert info["matcher_slots"] == 3, info        # maps of ~370 cells: three registrations in flight per matcher workgroup
    # the split leaves the builds just enough CUs for a whole number of rounds (one workgroup per map, four to a CU)
    n_cu = torch.cuda.get_device_properties(halls["dev"]).multi_processor_count
    whole = {n_cu - (-(-(-(-2 * B // (4 * k))) // 8) * 8) for k in range(1, 17)}
    assert info["matcher_groups"] in whole, (info, sorted(whole))
    for k, (T16, res) in enumerate(kept):
        r = same_bits(binding, T16, res, T_ref, r_ref, "call %d" % k)
    assert r["converged"].mean() > 0.9 and np.all(r["exit_code"] >= 0)
    # the maps the registrar holds are the maps of the plain build (slot of the last call: targets, then sources)
    last_slot = reg.mapset((26 - 1) % 8)
    nc = last_slot.num_cells_all()
    assert np.array_equal(nc[:B], r_ref["n_target"]) and np.array_equal(nc[B:], r_ref["n_source"])
    # ---- the oracle: 24 spread pairs + the 8 longest registrations of the batch
    T = kept[-1][0].cpu().numpy().reshape(B, 4, 4).transpose(0, 2, 1)
    sample = sorted(set(list(np.linspace(0, B - 1, 24).astype(int)) + list(np.argsort(-r_ref["fevals"])[:8])))
    assert len(sample) >= 28
    f_h, m_h = both[sample].cpu().numpy(), both[[B + b for b in sample]].cpu().numpy()
    for j, b in enumerate(sample):
        ot = O.OracleMap(RES, [0, 0, 0], SIZE); ot.load_points(f_h[j], RNG); ot.compute_cells()
        os_ = O.OracleMap(RES, [0, 0, 0], SIZE); os_.load_points(m_h[j], RNG); os_.compute_cells()
        if j < 4:
            for gpu, cpu in ((last_slot.export_cells(int(b)), ot.export_cells()), (last_slot.export_cells(int(B + b)), os_.export_cells())):
                assert np.array_equal(gpu[2], cpu[2]) and np.array_equal(gpu[3].astype(np.int64), cpu[3].astype(np.int64))
                assert np.max(np.abs(gpu[0] - cpu[0])) < 1e-9
        To, ro = O.match_d2d(ot, os_, halls["T_init"][b])
        dt = np.linalg.norm(T[b][:3, 3] - To[:3, 3])
        dr = 2.0 * np.arcsin(min(1.0, np.linalg.norm(T[b][:3, :3] - To[:3, :3]) / (2.0 * np.sqrt(2.0))))
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (b, dt, dr)
        assert bool(r["converged"][b]) == ro["converged"] and r["iterations"][b] == ro["iterations"], b
        assert abs(r["score"][b] - ro["score"]) < 1e-6 * abs(ro["score"])
    reg.close()


@pytest.mark.parametrize("fields", [{"matcher_groups": 96}, {"matcher_groups": 176, "build_streams": 1},
                                    {"matcher_form": 1}, {"matcher_slots": 2}, {"matcher_groups": 120, "matcher_slots": 3}])
def test_full_size_bits_do_not_depend_on_the_split(N, halls, fields):
    """the same calls on registrars whose parameters are FORCED away from the measured defaults (ndtgpu_registrar_params): 96
    or 176 matcher CUs, one build stream, one matcher launch per sub-batch, two or three registrations in flight per matcher
    workgroup -- the same bits"""
    import torch
    from ndt_feature_graph_amd import binding
    both, T0, T_ref, r_ref = halls["both"], halls["T0"], halls["T_ref"], halls["r_ref"]
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=B, depth=8, max_cells=4096, **fields)
    kept = run_calls(reg, torch, both, T0, n_calls=10, n_out=4)
    info = reg.info()
    if "matcher_groups" in fields:
        assert info["matcher_groups"] == fields["matcher_groups"] and info["calibrations"] == 0
    if fields.get("matcher_form") == 1:
        assert info["matcher_form"] == binding.MATCHER_PER_BATCH
    if "matcher_slots" in fields:
        assert info["matcher_slots"] == fields["matcher_slots"], info
    elif "matcher_groups" in fields:
        assert info["matcher_slots"] == 2, info       # a forced split is not measured: nothing is known about the maps
    for k, (T16, res) in enumerate(kept):
        same_bits(binding, T16, res, T_ref, r_ref, "call %d %r" % (k, fields))
    reg.close()


def test_one_registrar_halls_and_clutter_in_turn(N, O, halls):
    """ONE default registrar is fed the halls (1024 pairs, ~370 cells per map) and the cluttered scene (384 pairs, ~1700
    cells per map) in turn: every call returns the bits of the two-call path on its scene, and the registrar notices that
    its maps have changed -- it measures the split again (once or twice, not at every call)."""
    import torch
    from ndt_feature_graph_amd import binding, synth
    dev = halls["dev"]
    Bd = 384
    pr = synth.pair_2d(torch.arange(1, 1 + Bd, dtype=torch.int64, device=dev), NP, device=dev, chunk_bytes=1 << 30, scene="dense")
    dense = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
    pr["fixed"] = pr["moving"] = None
    T0d = pr["T_init"].transpose(1, 2).contiguous().reshape(Bd, 16)
    Td_ref, rd_ref = two_call(N, dense, T0d, Bd)
    assert rd_ref["n_target"].mean() > 3 * halls["r_ref"]["n_target"].mean()
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=B, max_cells=4096)
    st = torch.cuda.Stream(device=dev)
    n_calls = 40
    outs = []
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for k in range(n_calls):
            if k % 2 == 0:
                T16, res = halls["T0"].clone(), torch.zeros((B, 64), dtype=torch.uint8, device=dev)
                reg.submit(halls["both"][:B], halls["both"][B:], T16, res, range_limit=RNG, stream=st)
            else:
                T16, res = T0d.clone(), torch.zeros((Bd, 64), dtype=torch.uint8, device=dev)
                reg.submit(dense[:Bd], dense[Bd:], T16, res, range_limit=RNG, stream=st)
            outs.append((T16, res))
    reg.sync()
    torch.cuda.synchronize()
    for k, (T16, res) in enumerate(outs):
        if k % 2 == 0:
            same_bits(binding, T16, res, halls["T_ref"], halls["r_ref"], "halls, call %d" % k)
        else:
            same_bits(binding, T16, res, Td_ref, rd_ref, "dense, call %d" % k)
    info = reg.info()
    assert 2 <= info["calibrations"] <= 4, info
    assert info["matcher_slots"] in (2, 3) and info["resident_groups"] == 0, info      # (everything is done: no instance is left)
    # one dense pair against the oracle through this path (the scene has its own full test in tests/test_gpu_dense.py)
    T = outs[1][0].cpu().numpy().reshape(Bd, 4, 4).transpose(0, 2, 1)
    r = outs[1][1].cpu().numpy().view(binding.RESULT_DTYPE).reshape(Bd)
    b = int(np.argmax(rd_ref["fevals"]))
    ot = O.OracleMap(RES, [0, 0, 0], SIZE); ot.load_points(dense[b].cpu().numpy(), RNG); ot.compute_cells()
    os_ = O.OracleMap(RES, [0, 0, 0], SIZE); os_.load_points(dense[Bd + b].cpu().numpy(), RNG); os_.compute_cells()
    To, ro = O.match_d2d(ot, os_, pr["T_init"][b].cpu().numpy())
    dt = np.linalg.norm(T[b][:3, 3] - To[:3, 3])
    dr = 2.0 * np.arcsin(min(1.0, np.linalg.norm(T[b][:3, :3] - To[:3, :3]) / (2.0 * np.sqrt(2.0))))
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and r["iterations"][b] == ro["iterations"], (b, dt, dr)
    reg.close()


def test_wait_stream_refuses_the_matcher_priority(N, halls):
    """a caller's stream of the highest priority could share the matcher stream's hardware queue, where the device-side wait
    would sit in front of the instance it waits for: refused, not risked"""
    import torch
    dev = halls["dev"]
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=64, depth=4, max_cells=4096)
    both, T0 = halls["both"], halls["T0"]
    T16, res = T0[:64].clone(), torch.zeros((64, 64), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        t = reg.submit(both[:64], both[B:B + 64], T16, res, range_limit=RNG)
    hi = torch.cuda.Stream(device=dev, priority=-1)
    with pytest.raises(N.NdtGpuError):
        reg.wait_stream(hi, ticket=t)
    reg.wait_stream(torch.cuda.Stream(device=dev), ticket=t)          # default priority: fine
    reg.sync()
    reg.close()
