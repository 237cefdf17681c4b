#!/usr/bin/env python3
"""How well does "the last line search accepted its first trial" predict the next one?  Runs N bench pairs one at a time through
the host-driven Newton loop (NDTGPU_HOST_LOOP=1, NDTGPU_TRACE=1: one stderr line per evaluation) and replays the speculation
policies of csrc/ndt_solver.h offline: cost of an evaluation with / without the Hessian ~ 20.8 k / 9.3 k wave instructions.
usage: python tools/spec_stats.py [pairs=96] 2> trace.txt  (the script re-reads its own stderr through a pipe)"""
import os, subprocess, sys, re, collections
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import ndt_feature_graph_amd as N
    from ndt_feature_graph_amd import synth
    n = int(sys.argv[2])
    pr = synth.pair_2d(list(range(1, n + 1)), 100000)
    ms = N.MapSet(0.5, [0, 0, 0], [100, 100, 1], n_maps=2 * n, max_cells=4096)
    ms.build(np.concatenate([pr["fixed"].numpy(), pr["moving"].numpy()]), range_limit=30.0)
    os.environ["NDTGPU_HOST_LOOP"] = "1"; os.environ["NDTGPU_TRACE"] = "1"
    for k in range(n):
        sys.stderr.write("PAIR %d\n" % k); sys.stderr.flush()
        N.match_d2d(ms, k, ms, n + k, pr["T_init"][k].numpy())
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)], capture_output=True, text=True)
pairs, cur = [], None
for ln in p.stderr.splitlines():
    if ln.startswith("PAIR"):
        cur = []; pairs.append(cur)
    m = re.match(r"hip eval with_h (\d) phase (\d) itr (\d+) nfev (\d+)", ln)
    if m and cur is not None:
        cur.append(tuple(int(x) for x in m.groups()))
# a line search = the trial evaluations (phase 1) between two Newton evaluations; first trial accepted <=> exactly one trial
CH, CG = 20.8, 9.3
tot = collections.Counter()
seqs = []
for ev in pairs:
    ls, k = [], 0
    for (wh, ph, itr, nfev) in ev:
        if ph == 1:
            if nfev == 0 and k:          # (a trial with its Hessian that was accepted: the next search starts without a Newton evaluation)
                ls.append(k); k = 0
            k += 1
        elif k:
            ls.append(k); k = 0
    if k: ls.append(k)
    seqs.append(ls)
def cost(policy):
    c = 0.0
    for ls in seqs:
        last = False
        for i, k in enumerate(ls):
            ends = i == len(ls) - 1
            spec = policy(i, last) and not (ends and k == 1)      # (the shipped rule: no Hessian for a trial that ends the run)
            acc = k == 1
            c += (CH if spec else CG) + (k - 1) * CG               # the trials
            if not ends:
                c += 0.0 if (spec and acc) else CH                 # the next Newton evaluation
            last = acc
    return c
n_ls = sum(len(s) for s in seqs); n_acc = sum(sum(1 for k in s if k == 1) for s in seqs)
print("%d pairs, %d line searches, first trial accepted in %.1f %%; after an accepted one %.1f %%, after a rejected one %.1f %%" % (
    len(seqs), n_ls, 100.0 * n_acc / max(1, n_ls),
    100.0 * sum(sum(1 for a, b in zip(s, s[1:]) if a == 1 and b == 1) for s in seqs) / max(1, sum(sum(1 for a in s[:-1] if a == 1) for s in seqs)),
    100.0 * sum(sum(1 for a, b in zip(s, s[1:]) if a != 1 and b == 1) for s in seqs) / max(1, sum(sum(1 for a in s[:-1] if a != 1) for s in seqs))))
hist = collections.Counter(k for s_ in seqs for k in s_)
print("trials per line search:", dict(sorted(hist.items())))
for t in (2, 3, 4):
    reached = sum(v for k, v in hist.items() if k >= t); acc = hist.get(t, 0)
    print("  reached trial %d: %d, accepted there: %.1f %%" % (t, reached, 100.0 * acc / max(1, reached)))
base = cost(lambda i, last: False)
for name, pol in (("never", lambda i, last: False), ("last outcome (shipped)", lambda i, last: last), ("always", lambda i, last: True),
                  ("always after the first search", lambda i, last: i > 0), ("oracle", None)):
    if pol is None:
        c = 0.0
        for ls in seqs:
            for i, k in enumerate(ls):
                ends = i == len(ls) - 1
                spec = k == 1 and not ends
                c += (CH if spec else CG) + (k - 1) * CG + (0.0 if (spec or ends) else CH)
    else:
        c = cost(pol)
    print("  %-32s line-search + Newton evaluations: %.0f k instructions per pair (%.1f %% of never)" % (name, c / len(seqs), 100.0 * c / base))
