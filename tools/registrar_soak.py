#!/usr/bin/env python3
"""Soak of the registrar's stream-fed form: thousands of sub-batches of random sizes through small registrars (ring entries are
re-published every few hundred microseconds: the window in which a slot can hold a ticket of a batch that is gone), callers that
sometimes wait and sometimes do not, two registrars alive at once -- every call must return the bits of the two-call path.
usage (GPU box): python tools/registrar_soak.py [seconds=120]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ndt_feature_graph_amd as N
from ndt_feature_graph_amd import binding, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda", 0)
B, NP = 96, 4000
RES, SIZE, RNG = 0.5, [100.0, 100.0, 1.0], 30.0
pr = synth.pair_2d(torch.arange(8001, 8001 + B, dtype=torch.int64, device=dev), NP, device=dev)
both = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
T0 = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
DET = ["converged", "iterations", "fevals", "exit_code", "score", "n_source", "n_target", "pair_terms_g", "pair_terms_h"]
st = torch.cuda.current_stream()


def reference(per):
    """the two-call path cut into the registrar's launches: sub-batches of `per` pairs, each its own build launches"""
    T16 = T0.clone(); res = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    for off in range(0, B, per):
        p = min(per, B - off)
        ms = N.MapSet(RES, [0, 0, 0], SIZE, n_maps=2 * p, max_cells=4096)
        ms.build(both[off:off + p], range_limit=RNG, first=0, stream=st)
        ms.build(both[B + off:B + off + p], range_limit=RNG, first=p, stream=st)
        idx = torch.arange(p, dtype=torch.int32, device=dev)
        binding.match_batch_device(ms, idx, ms, idx + p, T16[off:off + p], res[off:off + p], p, stream=st)
        torch.cuda.synchronize()
    return T16.cpu().numpy(), res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)


rng = np.random.default_rng(3)
t_end = time.time() + budget
calls = subs = 0
rounds = 0
while time.time() < t_end:
    per = int(rng.choice([8, 12, 16, 24, 32, 48]))
    depth = int(rng.choice([2, 3, 4, 8]))
    T_ref, r_ref = reference(per)
    fields = {} if rng.random() < 0.7 else {"matcher_groups": int(rng.choice([8, 24, 64, 200]))}
    regs = [N.Registrar(RES, [0, 0, 0], SIZE, pairs_per_batch=per, depth=depth, max_cells=4096, **fields) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    n_out = int(rng.integers(2, 6))
    outs = [[(T0.clone(), torch.zeros((B, 64), dtype=torch.uint8, device=dev)) for _ in range(n_out)] for _ in regs]
    tickets = [[0] * n_out for _ in regs]
    kept = []
    torch.cuda.synchronize()
    for k in range(int(rng.integers(20, 60))):
        w = int(rng.integers(0, 2))
        reg, s = regs[w], streams[w]
        with torch.cuda.stream(s):
            T16, res = outs[w][k % n_out]
            if tickets[w][k % n_out]:
                reg.wait_stream(s, ticket=tickets[w][k % n_out])
                kept.append((T16.clone(), res.clone()))
            T16.copy_(T0)
            tickets[w][k % n_out] = reg.submit(both[:B], both[B:], T16, res, range_limit=RNG, stream=s)
        calls += 1; subs += (B + per - 1) // per
        if rng.random() < 0.15:
            reg.sync()
        if rng.random() < 0.05:
            time.sleep(0.002)
    for reg in regs:
        reg.sync()
    for w in range(2):
        kept += [(T16.clone(), res.clone()) for (T16, res), t in zip(outs[w], tickets[w]) if t]
    torch.cuda.synchronize()
    for T16, res in kept:
        assert np.array_equal(T16.cpu().numpy(), T_ref), "poses differ (per %d depth %d %r)" % (per, depth, fields)
        r = res.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
        for f in DET:
            assert np.array_equal(r[f], r_ref[f]), (f, per, depth, fields)
    for reg in regs:
        reg.close()
    rounds += 1
print("registrar soak: %d rounds, %d calls, %d sub-batches in %.0f s: every call the bits of the two-call path" % (rounds, calls, subs, budget))
