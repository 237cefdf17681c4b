"""ndtgpu_register_batch_device: scans in, poses out as ONE asynchronous C-ABI call (-m gpu).

The call shape of NDTFeatureGraph::updateLinksUsingNDTRegistration (ndt_feature_graph.cpp:347-353) and of the fuser's
loadPointCloud + computeNDTCells + match (ndt_feature_fuser_hmt.cpp:195-227, 353-357).  The registrar pipelines sub-batches over
its own streams; what it returns must be, bit for bit, what ndtgpu_mapset_build + ndtgpu_match_batch_device give for the same
scans, whatever the sub-batch size, the depth, the number of calls in flight and the streams the caller names -- and the oracle's
poses within the contract's tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DET_FIELDS = ["converged", "iterations", "fevals", "exit_code", "score", "n_source", "n_target", "pair_terms_g", "pair_terms_h"]
RES, SIZE, RNG = 0.5, [100.0, 100.0, 1.0], 30.0


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


@pytest.fixture(scope="module")
def scene(N):
    """96 scan pairs of 20 k points in HBM + the two-call reference result (build, then ndtgpu_match_batch_device)."""
    import torch
    from ndt_feature_graph_amd import binding, synth
    dev = torch.device("cuda", 0)
    B, NP = 96, 20000
    pr = synth.pair_2d(torch.arange(7001, 7001 + B, dtype=torch.int64, device=dev), NP, device=dev)
    both = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
    T0 = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
    ms = N.MapSet(RES, [0, 0, 0], SIZE, n_maps=2 * B, max_cells=4096)
    st = torch.cuda.current_stream()
    ms.build(both, range_limit=RNG, stream=st)
    idx = torch.arange(B, dtype=torch.int32, device=dev)
    T16 = T0.clone()
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    binding.match_batch_device(ms, idx, ms, idx + B, T16, res, B, stream=st)
    torch.cuda.synchronize()
    return {"B": B, "NP": NP, "both": both, "T0": T0, "T_ref": T16.cpu().numpy(), "T_init": pr["T_init"].cpu().numpy(),
            "r_ref": res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B), "dev": dev, "maps": ms}


def same_bits(binding, T16, res, T_ref, r_ref):
    assert np.array_equal(T16.cpu().numpy(), T_ref)
    r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(-1)
    for f in DET_FIELDS:
        assert np.array_equal(r[f], r_ref[f]), f
    return r


def two_call_reference(N, scene, per):
    """ndtgpu_mapset_build + ndtgpu_match_batch_device, the scans cut into the launches the registrar makes for sub-batches of
    `per` pairs (a map's last bits depend on the shape of the launch that built it -- which kernel, how many workgroups per
    map: csrc/ndt_build*.hip -- so "the same bits" means the same launches)."""
    import torch
    from ndt_feature_graph_amd import binding
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    if per >= B:
        return scene["T_ref"], scene["r_ref"]
    st = torch.cuda.current_stream()
    T16 = scene["T0"].clone()
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    for off in range(0, B, per):
        p = min(per, B - off)
        ms = N.MapSet(RES, [0, 0, 0], SIZE, n_maps=2 * p, max_cells=4096)
        ms.build(both[off:off + p], range_limit=RNG, first=0, stream=st)
        ms.build(both[B + off:B + off + p], range_limit=RNG, first=p, stream=st)
        idx = torch.arange(p, dtype=torch.int32, device=dev)
        binding.match_batch_device(ms, idx, ms, idx + p, T16[off:off + p], res[off:off + p], p, stream=st)
        torch.cuda.synchronize()
    return T16.cpu().numpy(), res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)


@pytest.mark.parametrize("per,depth", [(96, 1), (32, 3), (40, 2)])
def test_registrar_same_bits_as_build_plus_match(N, scene, per, depth):
    """one call; sub-batches of 96 / 32 / 40 (ragged last one) pairs over 1 / 3 / 2 internal map sets"""
    import torch
    from ndt_feature_graph_amd import binding
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    T_ref, r_ref = two_call_reference(N, scene, per)
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=per, depth=depth, max_cells=4096)
    T16 = scene["T0"].clone()
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    reg.submit(both[:B], both[B:], T16, res, range_limit=RNG, stream=torch.cuda.current_stream())
    reg.sync()
    r = same_bits(binding, T16, res, T_ref, r_ref)
    assert r["converged"].mean() > 0.8
    # whatever the cut, the poses are those of the one-launch reference to rounding (the maps differ in their last bits only)
    assert np.max(np.abs(T16.cpu().numpy() - scene["T_ref"])) < 1e-6
    # the maps the registrar built are the maps of the plain build (slot 0 holds the first sub-batch: targets, then sources)
    if depth == 1 and per >= B:
        first_slot = reg.mapset(0)
        for k in (0, B - 1, B, 2 * B - 1):
            a, b = first_slot.export_cells(k), scene["maps"].export_cells(k)
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
    reg.close()


def test_registrar_calls_in_flight_and_streams(N, scene):
    """four calls submitted back to back from two caller streams (inputs made on those streams just before), separate
    target / source buffers that are not adjacent, then wait_stream + a copy on a third stream"""
    import torch
    from ndt_feature_graph_amd import binding
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    T_ref, r_ref = two_call_reference(N, scene, 48)
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=48, depth=3, max_cells=4096)
    sa, sb, sc = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    outs, tickets = [], []
    for k, st in enumerate((sa, sb, sa, sb)):
        with torch.cuda.stream(st):
            tg = both[:B].clone()                         # the inputs are produced on the caller's stream ...
            pad = torch.empty(1 + k, dtype=torch.float32, device=dev)
            src = both[B:].clone()                        # ... and sources do not follow targets in memory
            T16 = scene["T0"].clone()
            res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
            tickets.append(reg.submit(tg, src, T16, res, range_limit=RNG, stream=st))
            outs.append((tg, src, pad, T16, res))
    assert tickets == [2, 4, 6, 8]                        # two sub-batches of 48 pairs per call
    # a ticket covers its call and the calls before it: the first two calls' outputs on a stream that waited for ticket 2
    reg.wait_stream(sc, ticket=tickets[1])
    with torch.cuda.stream(sc):
        copies = [(o[3].clone(), o[4].clone()) for o in outs[:2]]
    reg.wait_stream(sc)                                   # ... and everything submitted so far
    with torch.cuda.stream(sc):
        copies += [(o[3].clone(), o[4].clone()) for o in outs[2:]]
    sc.synchronize()
    with pytest.raises(N.NdtGpuError):
        reg.wait_stream(sc, ticket=99)                    # a ticket nobody was given
    for T16, res in copies:
        same_bits(binding, T16, res, T_ref, r_ref)
    reg.sync()
    reg.close()


def test_registrar_against_the_oracle(N, O, scene):
    """8 of the pairs against the CPU oracle: cells of both maps exact, pose within 1e-4 m / 1e-4 rad, same iterations"""
    import torch
    from ndt_feature_graph_amd import binding
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=B, depth=1, max_cells=4096)
    T16 = scene["T0"].clone()
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    reg.submit(both[:B], both[B:], T16, res, range_limit=RNG)
    reg.sync()
    T = T16.cpu().numpy().reshape(B, 4, 4).transpose(0, 2, 1)
    r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
    ms = reg.mapset(0)
    for b in np.linspace(0, B - 1, 8).astype(int):
        f, m = both[b].cpu().numpy(), both[B + b].cpu().numpy()
        ot = O.OracleMap(RES, [0, 0, 0], SIZE); ot.load_points(f, RNG); ot.compute_cells()
        os_ = O.OracleMap(RES, [0, 0, 0], SIZE); os_.load_points(m, RNG); os_.compute_cells()
        gi, ci = ms.export_cells(int(b))[2], ot.export_cells()[2]
        assert np.array_equal(gi, ci)
        assert np.array_equal(ms.export_cells(int(B + b))[2], os_.export_cells()[2])
        To, ro = O.match_d2d(ot, os_, scene["T_init"][b])
        dt = np.linalg.norm(T[b][:3, 3] - To[:3, 3])
        dr = 2.0 * np.arcsin(min(1.0, np.linalg.norm(T[b][:3, :3] - To[:3, :3]) / (2.0 * np.sqrt(2.0))))
        assert dt <= 1e-4 and dr <= 1e-4, (b, dt, dr)           # the tolerance BASELINE.json's north_star states
        assert r["iterations"][b] == ro["iterations"] and bool(r["converged"][b]) == ro["converged"]
    reg.close()


def test_registrar_rejects_bad_arguments(N, scene):
    import torch
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    with pytest.raises(N.NdtGpuError):
        N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=0, depth=3)
    with pytest.raises(N.NdtGpuError):
        N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=8, depth=0)
    with pytest.raises(N.NdtGpuError):                           # the stream-fed form cannot be had with one map set
        N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=8, depth=1, matcher_form=2)
    with pytest.raises(N.NdtGpuError):
        N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=8, depth=2, matcher_form=7)
    with pytest.raises(N.NdtGpuError):                           # two or three registrations per matcher workgroup, or 0 = decide
        N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=8, depth=2, matcher_slots=4)
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=8, depth=1, max_cells=4096)
    assert reg.info()["matcher_form"] == 1 and reg.info()["calibrations"] == 0
    reg.sync()                                                   # nothing submitted: returns at once
    assert reg.kernel_ms() == (0.0, 0.0, 0)
    reg.close()


def test_registrar_host_form(N, scene):
    """ndtgpu_register_batch_host: host clouds in, host poses out, sub-batches staged under the work of the one before -- the
    bits of the device form (same launches: sub-batches of 40 pairs, sources behind targets in the staging area)"""
    import torch
    from ndt_feature_graph_amd import binding
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    per = 40
    # reference: the device form on buffers laid out like the host form's staging (ONE build launch per sub-batch)
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=per, depth=2, max_cells=4096)
    T16 = scene["T0"].clone()
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for off in range(0, B, per):
        p = min(per, B - off)
        pack = torch.cat([both[off:off + p], both[B + off:B + off + p]]).contiguous()
        reg.submit(pack[:p], pack[p:], T16[off:off + p], res[off:off + p], range_limit=RNG)
        reg.sync()
    T_ref = T16.cpu().numpy().reshape(B, 4, 4).transpose(0, 2, 1)
    r_ref = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
    reg.close()
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=per, depth=2, max_cells=4096)
    scans = both.cpu().numpy()
    for _ in range(2):                                            # (again: the staging areas are reused)
        T, r = reg.register_host(scans[:B], scans[B:], scene["T_init"], range_limit=RNG)
        assert np.array_equal(T, T_ref)
        for f in DET_FIELDS:
            assert np.array_equal(r[f], r_ref[f]), f
    reg.close()


@pytest.mark.parametrize("mode,groups,slots", [("stream", None, 0), ("stream", 8, 0), ("stream", 8, 3), ("stream", None, 2),
                                               ("launch_per_batch", None, 0)])
def test_registrar_long_run_both_matcher_forms(N, scene, mode, groups, slots):
    """Twelve calls (36 sub-batches: many turns of the ring of four map sets) without a host wait, four output buffers in
    rotation guarded by tickets -- through the stream-fed matcher (one running instance serves batch after batch; its CU share
    measured on the first sub-batch, or forced to 8 workgroups) and through the form with one matcher launch per sub-batch
    (ndtgpu_registrar_params.matcher_form): every call returns the bits of ndtgpu_mapset_build + ndtgpu_match_batch_device."""
    import torch
    from ndt_feature_graph_amd import binding
    fields = {"matcher_form": binding.MATCHER_STREAM_FED if mode == "stream" else binding.MATCHER_PER_BATCH}
    if groups:
        fields["matcher_groups"] = groups
    if slots:
        fields["matcher_slots"] = slots                          # (three: hit lists of 640 entries per share instead of 1024)
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    T_ref, r_ref = two_call_reference(N, scene, 32)
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=32, depth=4, max_cells=4096, **fields)
    assert reg.info()["matcher_form"] == fields["matcher_form"]
    st = torch.cuda.Stream(device=dev)
    outs = [(scene["T0"].clone(), torch.zeros((B, 64), dtype=torch.uint8, device=dev)) for _ in range(4)]
    tickets = [0] * 4
    kept = []
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for k in range(12):
            T16, res = outs[k % 4]
            if tickets[k % 4]:
                reg.wait_stream(st, ticket=tickets[k % 4])       # the call that wrote this buffer last is complete ...
                kept.append((T16.clone(), res.clone()))           # ... its results are read on the stream that waited
            T16.copy_(scene["T0"])
            tickets[k % 4] = reg.submit(both[:B], both[B:], T16, res, range_limit=RNG, stream=st)
    reg.sync()
    kept += [(T16.clone(), res.clone()) for T16, res in outs]
    torch.cuda.synchronize()
    assert len(kept) == 12
    for T16, res in kept:
        same_bits(binding, T16, res, T_ref, r_ref)
    if mode == "stream":
        assert reg.info()["matcher_slots"] == (slots if slots else (2 if groups else 3)), reg.info()
    reg.close()


def test_registrar_stream_form_edges(N, scene):
    """The stream-fed form at its edges: maps that overflowed max_cells are refused in the loader (exit code -3, pose untouched:
    a batch whose registrations ALL end there still completes), single-pair calls, a ticket of the measuring first sub-batch, two
    registrars alive at once, destruction with work in flight."""
    import torch
    from ndt_feature_graph_amd import binding
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    T_ref, r_ref = two_call_reference(N, scene, 32)
    st = torch.cuda.current_stream()
    # (a) every map overflows
    small = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=32, depth=4, max_cells=128)
    T16 = scene["T0"].clone()
    res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(2):
        small.submit(both[:B], both[B:], T16, res, range_limit=RNG, stream=st)
    small.sync()
    r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
    assert np.all(r["exit_code"] == -3) and not r["converged"].any() and torch.equal(T16, scene["T0"])
    # (b) two registrars at once; single-pair calls; the first ticket
    ra = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=32, depth=4, max_cells=4096)
    rb = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=32, depth=2, max_cells=4096)
    Ta, Tb = scene["T0"].clone(), scene["T0"].clone()
    resa = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    resb = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = ra.submit(both[:32], both[B:B + 32], Ta[:32], resa[:32], range_limit=RNG, stream=st)      # the measuring sub-batch
    ra.wait_stream(st, ticket=t0)
    first = Ta[:32].clone()
    rb.submit(both[:B], both[B:], Tb, resb, range_limit=RNG, stream=st)
    ra.submit(both[32:B], both[B + 32:], Ta[32:], resa[32:], range_limit=RNG, stream=st)
    ra.sync(); rb.sync()
    assert np.array_equal(first.cpu().numpy(), T_ref[:32]), "first"
    same_bits(binding, Ta, resa, T_ref, r_ref)
    same_bits(binding, Tb, resb, T_ref, r_ref)
    one_T, one_r = scene["T0"][5:6].clone(), torch.zeros((1, 64), dtype=torch.uint8, device=dev)
    ra.submit(both[5:6], both[B + 5:B + 6], one_T, one_r, range_limit=RNG, stream=st)
    ra.sync()
    assert pose_dist(one_T.cpu().numpy()[0], T_ref[5]) < 1e-6        # (a map built alone: another launch shape, the same pose)
    # (c) destroyed with work in flight: destruction waits
    Tb.copy_(scene["T0"])
    rb.submit(both[:B], both[B:], Tb, resb, range_limit=RNG, stream=st)
    rb.close(); ra.close(); small.close()
    torch.cuda.synchronize()
    same_bits(binding, Tb, resb, T_ref, r_ref)


def test_registrar_gives_up_cleanly_and_works_again(N, scene):
    """The give-up protocol of the stream-fed matcher, reached through the test aid ndtgpu_registrar_inject_abort:
    registrations that did not run report exit_code -4 with their pose untouched, every other one carries the bits of the
    two-call path, ndtgpu_registrar_sync reports the abort ONCE, and the same registrar then registers everything again."""
    import torch
    from ndt_feature_graph_amd import binding
    B, both, dev = scene["B"], scene["both"], scene["dev"]
    T_ref, r_ref = two_call_reference(N, scene, 32)
    reg = N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=32, depth=4, max_cells=4096, matcher_form=binding.MATCHER_STREAM_FED)
    st = torch.cuda.current_stream()
    T0 = scene["T0"]
    outs = [(T0.clone(), torch.zeros((B, 64), dtype=torch.uint8, device=dev)) for _ in range(6)]
    torch.cuda.synchronize()
    for T16, res in outs[:1]:                                   # (the measuring sub-batch and two more run to the end)
        reg.submit(both[:B], both[B:], T16, res, range_limit=RNG, stream=st)
    reg.sync()
    for T16, res in outs[1:]:
        reg.submit(both[:B], both[B:], T16, res, range_limit=RNG, stream=st)
    reg.inject_abort()
    with pytest.raises(N.NdtGpuError):
        reg.sync()
    torch.cuda.synchronize()
    same_bits(binding, outs[0][0], outs[0][1], T_ref, r_ref)
    not_run = 0
    for T16, res in outs[1:]:
        r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
        T = T16.cpu().numpy()
        skipped = r["exit_code"] == -4
        not_run += int(skipped.sum())
        assert not r["converged"][skipped].any() and np.array_equal(T[skipped], T0.cpu().numpy()[skipped])
        assert np.array_equal(T[~skipped], T_ref[~skipped])
        for f in DET_FIELDS:
            assert np.array_equal(r[f][~skipped], r_ref[f][~skipped]), f
    assert not_run > 0, "the abort came after everything had run: nothing was tested"
    reg.sync()                                                  # reported once; the registrar is usable again
    T16, res = T0.clone(), torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    for _ in range(3):
        T16.copy_(T0)
        reg.submit(both[:B], both[B:], T16, res, range_limit=RNG, stream=st)
        reg.sync()
        same_bits(binding, T16, res, T_ref, r_ref)
    reg.close()


def test_registrar_soak_short(N):
    """tools/registrar_soak.py for 12 s: hundreds of calls of random sizes through small registrars (ring entries re-published every
    few hundred microseconds), two registrars alive at once, callers that sometimes wait: every call the two-call path's bits.
    (The long form, 180 s on one MI355X: 48 838 calls, 284 705 sub-batches, no difference, no stall -- profiles/README.md.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "registrar_soak.py"), "12"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "every call the bits of the two-call path" in out.stdout


def pose_dist(a16, b16):
    return float(np.max(np.abs(np.asarray(a16) - np.asarray(b16))))
