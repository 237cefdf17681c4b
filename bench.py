#!/usr/bin/env python3
"""bench.py -- NDT scan-pair registrations/sec (100k pts, 0.5 m cells) on MI355X.

One STEP = one pass of the hot path over one batch of synthetic scan pairs resident in HBM:
    grid build (K1-K3) of the 2*B scans  +  D2D match (K4-K5) of the B pairs to convergence
(+ the RCCL all-gather of the edge transforms when WORLD_SIZE > 1).  Workload = BASELINE.json
configs[2] ("batch of 1024 independent scan pairs, 100k pts each, 1xMI355X"), the single-GPU
configuration the throughput metric is quoted on; per rank the batch is fixed (weak scaling).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (largest share of the step),
`kernels` lists both; `cpu_baseline` times the CPU oracle on a bounded sample of the same pairs through a C driver
(oracle/cpu_baseline.c: -O3 -march=native, taskset-pinned, warm-up + median of 5 passes, no Python in the timed loop;
plus the labelled OpenMP all-cores figure) and `parity` compares the GPU poses of that sample with the oracle's.

  python bench.py --config 4 [--nodes N] [--gated]     configs[3]: replay harness -- N node maps, all-pairs (or gated)
                                                       candidate edges sharded over the ranks, nodes/s, edges/s and the
                                                       final all-gather reported separately
  python bench.py --config 5                           configs[4]: 3D mode -- one 200 k-point pair (latency) and a batch of
                                                       64 sweeps / 32 pairs (throughput), roofline of build and matcher
  python bench.py --config fuse                        the node-map path: add_cloud of 256 node maps (ray
                                                       tracing + accumulate + finalise), then matcher, covariance and
                                                       occupancy overlap of the 19 900 links of 200 fused maps
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
ROUND_TAG = "r06"
PMC_FILE = ROUND_TAG + "_pmc_traffic.json"


def cpu_info():
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline_c(fixed_h, moving_h, T0, res, size_m, rng_lim, delta, nn, itr_max, reps=5, tag="bench"):
    """Times oracle/cpu_baseline.c (built here with -O3 -march=native) on the given sample: one pinned thread, and all
    cores with OpenMP.  Returns (single dict, omp dict, poses [S,4,4])."""
    import subprocess
    import tempfile
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-s", "-C", odir, "_bin/cpu_baseline", "_bin/cpu_baseline_omp"])
    S, NPts = fixed_h.shape[0], fixed_h.shape[1]
    path = os.path.join(tempfile.gettempdir(), "ndt_%s_%d.bin" % (tag, os.getpid()))
    with open(path, "wb") as f:
        f.write(np.array([S, NPts], dtype=np.int32).tobytes())
        f.write(np.array([res] + list(size_m) + [rng_lim, delta], dtype=np.float64).tobytes())
        f.write(np.array([nn, itr_max], dtype=np.int32).tobytes())
        for b in range(S):
            f.write(np.ascontiguousarray(fixed_h[b], dtype=np.float32).tobytes())
            f.write(np.ascontiguousarray(moving_h[b], dtype=np.float32).tobytes())
            f.write(np.ascontiguousarray(T0[b].T, dtype=np.float64).tobytes())      # column-major
    cores = sorted(os.sched_getaffinity(0))
    pin = cores[len(cores) // 2]
    one = json.loads(subprocess.check_output(["taskset", "-c", str(pin), os.path.join(odir, "_bin", "cpu_baseline"), path,
                                              str(reps), "1"]).decode().strip().splitlines()[-1])
    poses = np.fromfile(path + ".out", dtype=np.float64).reshape(S, 4, 4).transpose(0, 2, 1).copy()
    nthr = min(len(cores), S)
    omp = json.loads(subprocess.check_output([os.path.join(odir, "_bin", "cpu_baseline_omp"), path, "3",
                                              str(nthr)]).decode().strip().splitlines()[-1])
    for p_ in (path, path + ".out"):
        try:
            os.remove(p_)
        except OSError:
            pass
    one["pinned_core"] = pin
    return one, omp, poses


def config4(args, torch, dist, N, binding, synth, rank, world, dev, size_m, rng_lim):
    """BASELINE configs[3]: "ndt_offline_ndt_feature bag replay, ~5k nodes, all-pairs candidate edges sharded over 8 GPUs".
    Synthetic replay (the bags are not in the reference): a trajectory with one node every 2 m (newNodeTranslDist,
    ndt_graph_offline.cpp:301) through a floor plan of separate rooms (100 nodes per room on a serpentine path), every
    node a FUSED node map -- `--scans-per-node` scans taken while the vehicle moves through the node's first metre, added one
    after the other with ndtgpu_mapset_add_cloud (ray tracing + accumulate + finalise: what graph.cpp:273 really matches).
    One step, as SURVEY.md 8(e) lays it out:
      A. node maps built data-parallel: node k on rank k mod world (its scans exist on that rank only);
      B. ONE all-gather of the packed cell records (cells + occupancies; ndtgpu_mapset_pack_cells_device / _unpack): every
         rank then holds every node map;
      C. the candidate edges, dealt block-cyclically (chunk 256), registered on the node maps with the edge preset;
      D. the final all-gather of the edge results.
    The candidate edges that are TIMED (and are `value`) are the GATED ones: pairs whose odometry poses lie within
    --gate-dist of each other and at least two indices apart -- the stand-in for the reference's FLIRT candidate matching +
    getValidLinks gates (ndt_feature_graph_opt.cpp:49-52, :131-160).  All 12 497 500 pairs of computeAllPossibleLinks
    (ndt_feature_graph.cpp:395-405) are registered once more outside the timed region and reported as a labelled extra:
    98 % of them join rooms that share nothing and end at their first evaluation, so their rate says little.
    --all-pairs makes them the timed workload instead.  For the gated edges of rank 0 the per-link covariance and occupancy
    overlap of updateLinksUsingNDTRegistration (graph.cpp:296-340) are timed too."""
    from ndt_feature_graph_amd import distributed as D
    n_nodes, res = args.nodes, args.res
    S, NPn, per_room = args.scans_per_node, args.node_points, 100
    q = np.arange(n_nodes) % per_room
    room = np.arange(n_nodes) // per_room
    col, row = q // 10, q % 10
    row = np.where(col % 2 == 1, 9 - row, row)                       # serpentine: 2 m between consecutive nodes
    lx, ly = -9.0 + 2.0 * col, -9.0 + 2.0 * row
    yaw = np.where(col % 2 == 1, -np.pi / 2, np.pi / 2)
    local = np.stack([lx, ly, yaw], axis=1)
    world_xy = np.stack([80.0 * (room % 8) + lx, 80.0 * (room // 8) + ly], axis=1)
    node_T = synth.pose2d_to_T(torch.as_tensor(np.concatenate([world_xy, yaw[:, None]], axis=1))).numpy()
    g = np.random.default_rng(11)
    odo_T = node_T.copy()
    odo_T[:, 0, 3] += g.normal(scale=0.03, size=n_nodes)
    odo_T[:, 1, 3] += g.normal(scale=0.03, size=n_nodes)
    # phase A input: the scans of THIS rank's nodes, in the node frame (the vehicle drives 0.9 m / S per scan straight ahead)
    my_nodes = D.shard_nodes(n_nodes, rank, world)
    n_local = len(my_nodes)
    seeds = torch.as_tensor(4000 + room[my_nodes], dtype=torch.int64, device=dev)
    clouds = []
    for k in range(S):
        dx = 0.9 * k / max(1, S)
        pk = local[my_nodes].copy()
        pk[:, 0] += dx * np.cos(yaw[my_nodes]); pk[:, 1] += dx * np.sin(yaw[my_nodes])
        sc = synth.scan_2d(seeds, torch.as_tensor(pk, device=dev), NPn, noise_stream=k, chunk_bytes=2 << 30).contiguous()
        sc[:, :, 0] += dx                                              # sensor frame -> node frame
        clouds.append((sc, np.tile(np.array([[dx, 0.0, 0.0]]), (n_local, 1))))
    all_edges = D.all_pairs(n_nodes)
    d_odo = np.linalg.norm(odo_T[all_edges[:, 0], :2, 3] - odo_T[all_edges[:, 1], :2, 3], axis=1)
    gate_all = (d_odo <= args.gate_dist) & ((all_edges[:, 1] - all_edges[:, 0]) >= 2)
    n_gated = int(gate_all.sum())
    timed_all_pairs = bool(args.all_pairs)
    edges = all_edges if timed_all_pairs else all_edges[gate_all]
    gate = gate_all if timed_all_pairs else np.ones(len(edges), dtype=bool)
    n_edges = len(edges)
    st = torch.cuda.current_stream()
    CH = 1 << 20
    local_set = N.MapSet(res, [0, 0, 0], size_m, n_maps=max(1, n_local), max_cells=4096)
    local_set.enable_occupancy()
    pool = N.MapSet(res, [0, 0, 0], size_m, n_maps=n_nodes, max_cells=4096)
    pool.enable_occupancy()

    def build_local():
        local_set.clear()
        for k, (sc, org) in enumerate(clouds):
            local_set.add_cloud(sc, org, stream=st, **(dict(maxz=100.0, sensor_noise=0.1) if k == 0 else dict(maxz=25.0, sensor_noise=0.06)))

    # the record size is a configuration value like max_cells: the largest node map of this replay, rounded up (one untimed
    # build; a map with more cells would be cut and flagged, and refused by the matcher like an overflowing build)
    build_local()
    cap_local = int(local_set.num_cells_all()[:n_local].max()) if n_local else 0
    occ_local = local_set.occupied_cells_max(0, n_local, stream=st) if n_local else 0
    capt = torch.tensor([cap_local, occ_local], dtype=torch.int64, device=dev)
    if args.use_dist:
        dist.all_reduce(capt, op=dist.ReduceOp.MAX)
    cells_cap = min(4096, (int(capt[0].item()) * 5 // 4 + 63) // 64 * 64)
    # occupancies travel as (slot, value) pairs of the cells that have a reading (2-3 % of a node map's 80 000 slots)
    occ_cap = (int(capt[1].item()) * 5 // 4 + 63) // 64 * 64
    stride = local_set.pack_bytes(cells_cap, occ_cap=occ_cap)
    stride_dense = local_set.pack_bytes(cells_cap, True)
    packed = torch.zeros((max(1, (n_nodes + world - 1) // world), stride), dtype=torch.uint8, device=dev)

    class Edges:
        def __init__(self, e):
            self.e = e
            self.mine = D.shard_edges(len(e), rank, world, 256)
            T0 = np.einsum("eij,ejk->eik", np.linalg.inv(odo_T)[e[self.mine, 0]], odo_T[e[self.mine, 1]])
            self.T0_cm = torch.as_tensor(np.ascontiguousarray(T0.transpose(0, 2, 1)).reshape(-1, 16), device=dev)
            self.ti = torch.as_tensor(e[self.mine, 0].astype(np.int32), device=dev)
            self.si = torch.as_tensor(e[self.mine, 1].astype(np.int32), device=dev)
            self.T16 = self.T0_cm.clone()
            self.results = torch.zeros((len(self.mine), 64), dtype=torch.uint8, device=dev)

    def barrier():
        if args.use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass(E, rebuild=True):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record(st)
        if rebuild:
            build_local()                                                            # phase A: this rank's node maps
        ev[1].record(st)
        if rebuild:
            local_set.pack_cells(packed, 0, n_local, cells_cap=cells_cap, occ_cap=occ_cap, stream=st)       # phase B
            allrec = D.exchange_node_maps(packed[:n_local], n_nodes, rank, world)
            pool.unpack_cells(allrec, 0, n_nodes, with_occupancy=True, stream=st)
        ev[2].record(st)

        def register(my_edges):                                                      # phase C: this rank's edges
            assert len(my_edges) == len(E.mine)
            E.T16.copy_(E.T0_cm)
            for c0 in range(0, len(E.mine), CH):
                c1 = min(len(E.mine), c0 + CH)
                binding.match_batch_device(pool, E.ti[c0:c1], pool, E.si[c0:c1], E.T16[c0:c1], E.results[c0:c1], c1 - c0,
                                           stream=st, delta_score=1e-3)
            ev[3].record(st)
            return E.T16, E.results
        _, Tg_, Rg_ = D.register_sharded(len(E.e), rank, world, register, 256)       # phase D inside: the only collective
        ev[4].record(st)
        return ev, (Tg_, Rg_)

    E = Edges(edges)
    steps = args.steps if args.steps != 100 else 3
    barrier()
    for _ in range(1 if args.warmup == 5 else max(1, args.warmup)):
        one_pass(E)
    barrier()
    t0 = time.perf_counter()
    evs = []
    for _ in range(steps):
        ev, gathered = one_pass(E)
        evs.append(ev)
    barrier()
    elapsed = time.perf_counter() - t0
    if args.use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    build_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    exch_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    match_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))
    gather_ms = float(np.mean([e[3].elapsed_time(e[4]) for e in evs]))
    r = gathered[1].cpu().numpy().view(binding.RESULT_DTYPE).reshape(-1)
    Tg = gathered[0].cpu().numpy().reshape(-1, 4, 4).transpose(0, 2, 1)
    cells = pool.num_cells_all()
    # the gated edges: result quality + covariance + overlap (rank 0, host arrays like the reference's call sites)
    gi = np.nonzero(gate)[0]
    extra = {}
    if rank == 0 and len(gi):
        rel_true = np.einsum("eij,ejk->eik", np.linalg.inv(node_T)[edges[gi, 0]], node_T[edges[gi, 1]])
        err = np.linalg.norm(Tg[gi][:, :2, 3] - rel_true[:, :2, 3], axis=1)
        same_room = room[edges[gi, 0]] == room[edges[gi, 1]]
        sub = gi[:200000]
        tcov = _timed(torch, lambda: binding.covariance(pool, edges[sub, 0], pool, edges[sub, 1], Tg[sub]), reps=1, warm=1)
        tovl = _timed(torch, lambda: binding.overlap_score(pool, edges[sub, 0], pool, edges[sub, 1], Tg[sub]), reps=1, warm=1)
        # what the reference does with the registered links next: getValidLinks (ndt_feature_graph.cpp:527-556) with the
        # defaults of ndt_feature_graph_opt.cpp:49-52 -- score <= 0.1, >= 2 indices apart, the moving node's pose predicted
        # through the link within 1.0 m / 0.2 rad of its odometry pose
        sc_sub, _ = binding.overlap_score(pool, edges[sub, 0], pool, edges[sub, 1], Tg[sub])
        kept = D.valid_links(edges[sub], Tg[sub], odo_T, scores=sc_sub)
        kept_geo = D.valid_links(edges[sub], Tg[sub], odo_T)
        extra = {"gated_edges": int(len(gi)), "gated_same_room_frac": float(same_room.mean()),
                 "reference_getValidLinks": {"max_score": 0.1, "max_dist_m": 1.0, "max_angular_dist_rad": 0.2, "min_idx_dist": 2,
                                             "links_in": int(len(sub)), "links_kept": int(len(kept)),
                                             "links_kept_without_the_score_test": int(len(kept_geo)),
                                             "note": "the reference's 1.0 m / 0.2 rad / 2 are a filter on REGISTERED links "
                                                     "(ndt_feature_graph_opt.cpp:155), not a candidate gate; candidates come "
                                                     "from FLIRT matching (not built), for which --gate-dist stands in"},
                 "gated_converged_frac": float(r["converged"][gi].mean()), "gated_mean_iterations": float(r["iterations"][gi].mean()),
                 "gated_median_translation_error_m": float(np.median(err)), "gated_err_below_5cm_frac": float((err < 0.05).mean()),
                 "covariance_us_per_edge": 1e3 * tcov / len(sub), "overlap_us_per_edge": 1e3 * tovl / len(sub),
                 "covariance_overlap_sample": int(len(sub))}
    hist = np.bincount(np.minimum(r["iterations"], 32), minlength=33).tolist()
    # the labelled extra: all pairs once (node maps as they stand), outside the timed region
    all_extra = None
    if not timed_all_pairs and not args.no_all_pairs:
        EA = Edges(all_edges)
        one_pass(EA, rebuild=False); barrier()
        tA = time.perf_counter()
        evA, gA = one_pass(EA, rebuild=False)
        barrier()
        tA = time.perf_counter() - tA
        rA = gA[1].cpu().numpy().view(binding.RESULT_DTYPE).reshape(-1)
        all_extra = {"edges": int(len(all_edges)), "match_and_gather_ms": 1e3 * tA,
                     "edge_registrations_per_s": len(all_edges) / tA,
                     "edges_without_any_pair_term": int((rA["pair_terms_h"] == 0).sum()),
                     "edges_with_pair_terms": int((rA["pair_terms_h"] > 0).sum()),
                     "note": "all 12.5 M pairs of computeAllPossibleLinks on the node maps of the last step; an edge between rooms "
                             "that share no cell ends at its first evaluation -- this rate is not comparable with `value`"}
        del EA
    what = "all-pairs" if timed_all_pairs else "gated"
    # (the exchange and the final gather as they came back, against what this rank put in: its own records / rows)
    coll = {"process_group": "nccl (RCCL)" if args.use_dist else None, "world": world,
            "forced_on_one_rank": bool(args.use_dist and world == 1),
            "collectives_per_step": {"exchange_all_gather": 1, "edge_result_all_gathers": 2} if args.use_dist else {},
            "gathered_rows_equal_local": (bool(torch.equal(gathered[0][torch.as_tensor(E.mine, device=dev)], E.T16))
                                          if args.use_dist else None)}
    model = D.phase_model(build_ms * world, match_ms * world, n_nodes, stride, n_edges) if world == 1 else None
    out = {"metric": "NDT graph-edge registrations/sec on fused node maps (%d scans x %d pts per node, %.2f m cells), %s candidate edges"
                     % (S, NPn, res, what),
           "value": n_edges * steps / elapsed,
           "unit": "registrations/s", "n_gpus": world, "steps": steps, "warmup": 1, "ms_per_step": 1e3 * elapsed / steps,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "configs[3]: graph replay, %d nodes 2 m apart through %d rooms, every node a fused map of %d scans x %d pts "
                                  "(%.2f m cells); one step = A. build the node maps k %% world == rank (%d add_cloud calls), B. pack + ONE "
                                  "all-gather + unpack of the cell records (%d B per node: %d cells + the (slot, occupancy) pairs of up to %d cells with a reading), C. register this rank's "
                                  "block-cyclic share (chunk 256) of the %d %s candidate edges with the edge preset (DELTA_SCORE 1e-3), "
                                  "D. all-gather of the edge results; %d rank(s)" % (
                                      n_nodes, int(room.max()) + 1, S, NPn, res, S, stride, cells_cap, occ_cap, n_edges, what, world),
                      "nodes": n_nodes, "nodes_this_rank": int(n_local), "edges": n_edges, "edges_this_rank": int(len(E.mine)),
                      "edges_within_gate": n_gated, "gate_dist_m": args.gate_dist, "mean_cells_per_node_map": float(cells.mean()),
                      "record_bytes_per_node": int(stride), "cells_cap": int(cells_cap), "occ_cap": int(occ_cap),
                      "record_bytes_per_node_dense_occupancy": int(stride_dense)},
           "nodes_per_s_build": n_nodes / (build_ms * 1e-3) if world == 1 else n_local * world / (build_ms * 1e-3),
           "scans_per_s_fused": n_local * world * S / (build_ms * 1e-3),
           "edges_per_s_match": len(E.mine) / (match_ms * 1e-3) * world,
           "phase_ms": {"A_build_my_nodes": build_ms, "B_pack_allgather_unpack": exch_ms, "C_match_my_edges": match_ms,
                        "D_all_gather_edge_results": gather_ms},
           "phase_model_1_to_8_gpus": model,
           "phase_model_note": "from this one-rank run: builds and registrations divide by the world size, the exchange is an all-gather "
                               "of the node records over point-to-point xGMI (7 links x ~153 GB/s per GPU); a model, no 8-GPU node "
                               "was available (distributed.phase_model)",
           "gather_bytes_per_edge": 16 * 8 + 64, "converged_frac": float(r["converged"].mean()),
           "mean_iterations": float(r["iterations"].mean()), "iteration_histogram_0_to_32": hist,
           "edges_without_any_pair_term": int((r["pair_terms_h"] == 0).sum()), "gated": extra, "all_pairs_extra": all_extra,
           "note": "value counts registrations of the %s candidate edges on fused node maps (the unit the graph layer consumes, "
                   "graph.cpp:273) per second of whole steps (node builds + exchange + registrations + gather); a node map is built "
                   "once per node per step, not per edge" % what}
    out["collectives"] = coll
    if rank == 0:
        print(json.dumps(out))
    if args.use_dist:
        dist.destroy_process_group()


def fuser_leg(torch, N, binding, synth, dev, with_cpu, n_fusers=64, n_points=100000, n_updates=6):
    """NDTFeatureFuserHMT::update (ndt_feature_fuser_hmt.cpp:108-512) as the fuser bank runs it: ONE ndtgpu_fuser_update_batch
    call per scan step for `n_fusers` independent fusers (scan -> scan map on the node map's lattice -> matchFusion with the
    soft constraint, the Tikhonov term and the 40 odometry cells -> covariance -> pose -> ray-traced fuse-in), 100 k points per
    scan, 0.5 m cells, the fuser preset; beside it one fuser alone (the reference's call shape) and the CPU oracle walking one
    fuser's update.  A labelled extra of the bench line, outside the timed region."""
    import math
    res, rng_lim, size = 0.5, 30.0, [100.0, 100.0, 1.0]
    prm = N.fuser_params(resolution=res, map_size_x=size[0], map_size_y=size[1], map_size_z=size[2], sensor_range=rng_lim,
                         neighbours=2, itr_max=30, delta_score=1e-6, max_cells=4096)
    gen = np.random.default_rng(11)
    seeds = torch.arange(9001, 9001 + n_fusers, dtype=torch.int64, device=dev)
    poses = np.zeros((n_updates + 1, n_fusers, 3))
    for s in range(1, n_updates + 1):
        step = np.stack([gen.uniform(0.15, 0.3, n_fusers), gen.uniform(-0.05, 0.05, n_fusers), gen.uniform(-0.04, 0.04, n_fusers)], axis=1)
        c, sn = np.cos(poses[s - 1, :, 2]), np.sin(poses[s - 1, :, 2])
        poses[s, :, 0] = poses[s - 1, :, 0] + c * step[:, 0] - sn * step[:, 1]
        poses[s, :, 1] = poses[s - 1, :, 1] + sn * step[:, 0] + c * step[:, 1]
        poses[s, :, 2] = poses[s - 1, :, 2] + step[:, 2]
    scans = [synth.scan_2d(seeds, torch.as_tensor(poses[s], device=dev), n_points, noise_stream=s).contiguous() for s in range(n_updates + 1)]

    def T2(p):
        T = np.tile(np.eye(4), (p.shape[0], 1, 1))
        T[:, 0, 0] = np.cos(p[:, 2]); T[:, 0, 1] = -np.sin(p[:, 2]); T[:, 1, 0] = np.sin(p[:, 2]); T[:, 1, 1] = np.cos(p[:, 2])
        T[:, 0, 3], T[:, 1, 3] = p[:, 0], p[:, 1]
        return T
    Tm = []
    for s in range(n_updates):
        true = np.linalg.inv(T2(poses[s])) @ T2(poses[s + 1])
        noise = T2(np.stack([gen.normal(0, 0.01, n_fusers), gen.normal(0, 0.01, n_fusers), gen.normal(0, 0.003, n_fusers)], axis=1))
        Tm.append(true @ noise)
    st = torch.cuda.current_stream()

    def run(count):
        bank = N.FuserBank(prm, count)
        bank.initialize(T2(poses[0][:count]), scans[0][:count], stream=st)
        bank.update(Tm[0][:count], scans[1][:count], stream=st)         # (warm: code objects, staging buffers)
        bank.poses()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(1, n_updates):
            bank.update(Tm[s][:count], scans[s + 1][:count], stream=st)
            T, r = bank.poses()                                         # the pose is what update() returns: fetched every step
        dt = (time.perf_counter() - t0) / (n_updates - 1)
        bank.close()
        return dt, T, r
    dt_all, T_all, r_all = run(n_fusers)
    dt_one, T_one, r_one = run(1)
    true_last = T2(poses[n_updates])
    err = float(np.max(np.linalg.norm(T_all[:, :2, 3] - true_last[:, :2, 3], axis=1)))
    out = {"unit": "fuser updates/s", "value": n_fusers / dt_all, "fusers_per_call": n_fusers, "ms_per_call": 1e3 * dt_all,
           "one_fuser_ms_per_update": 1e3 * dt_one, "one_fuser_equals_batch_bits": bool(np.array_equal(T_one[0], T_all[0])),
           "worst_position_error_vs_ground_truth_m": err, "matcher_iterations_mean": float(r_all["match"]["iterations"].mean()),
           "workload": "%d independent fusers x %d points per scan, 0.5 m cells, node maps 100 x 100 x 1 m, default Params() of the fuser "
                       "(useOdom, useSoftConstraints, useTikhonovRegularization, computeCov) with the fuser preset of the matcher; "
                       "ONE ndtgpu_fuser_update_batch call + ndtgpu_fuser_poses per scan step (host wall clock, poses fetched every "
                       "step like update() returns them)" % (n_fusers, n_points)}
    if with_cpu:
        import oracle as O
        # the CPU oracle walks ONE update of fuser 0 from the same state (node map after initialize + first update is rebuilt there)
        k = 0
        sens = np.eye(4)
        Tn = T2(poses[0][:1])[0]
        om = O.OracleMap(res, [Tn[0, 3], Tn[1, 3], 0.0], size)

        def mv(T, xyz):
            x, y, z = (xyz[:, a].astype(np.float64) for a in range(3))
            return np.stack([(T[r, 0] * x + T[r, 1] * y + T[r, 2] * z + T[r, 3]).astype(np.float32) for r in range(3)], axis=1)
        sc0 = scans[0][k].cpu().numpy()
        om.add_point_cloud((Tn @ sens)[:3, 3], mv(Tn, sc0), maxz=100.0, sensor_noise=0.1, order_free=True)
        om.compute_cells_full()
        sc1 = scans[1][k].cpu().numpy()
        c0 = time.perf_counter()
        pp = N.fuser_prepare(prm, Tn, Tm[0][k], [Tn[0, 3], Tn[1, 3], 0.0])
        local = rng_lim + 3 * res
        os_ = O.OracleMap(res, pp["scan_centre"], [local, local, size[2]])
        os_.load_points(mv(pp["Tscan"].reshape(4, 4).T, sc1), rng_lim, range_origin=pp["range_origin"])
        os_.compute_cells()
        sm = np.tile(pp["feat_src_mean"], (40, 1)); tm = np.tile(pp["feat_tgt_mean"], (40, 1))
        scv = np.tile(pp["feat_cov_rotated"], (40, 1)); scv[39] = pp["feat_cov_plain"]; tcv = np.tile(pp["feat_cov_rotated"], (40, 1))
        To, ro = O.match_fusion_feat(om, os_, Tm[0][k], pp["Tcov"].reshape(6, 6), (sm, scv, tm, tcv), use_soft_constraints=True, tikhonov=True,
                                     step_control_fusion=True, n_neighbours=2, itr_max=30, delta_score=1e-6, step_control=1)
        O.covariance(om, os_, To, n_neighbours=2)
        Tn2 = Tn @ (To if ro["converged"] else Tm[0][k])
        om.add_point_cloud((Tn2 @ sens)[:3, 3], mv(Tn2 @ sens, sc1), maxz=25.0, sensor_noise=0.06, order_free=True)
        om.compute_cells_full()
        t_cpu = time.perf_counter() - c0
        out["cpu_oracle_one_update_ms"] = 1e3 * t_cpu
        out["one_fuser_speedup_vs_cpu_1thread"] = t_cpu / dt_one
        out["batch_speedup_vs_cpu_1thread"] = t_cpu * n_fusers / dt_all
    return out


def dense_scene_leg(args, torch, N, binding, synth, dev, size_m, rng_lim, with_cpu):
    """The headline configuration (100 k points, 0.5 m cells, fuser preset) on the DENSE scene of synth.room_2d -- ~2 k Gaussian
    cells per map, the size SURVEY.md 8(a, d) gives a 2D map, against ~370 in the plain room the headline runs on: the same two
    launches per step (all maps, all registrations), serial on one stream, fewer pairs (the scene generator is the slow
    part).  Returns the labelled `dense_scene` object: registrations/s, kernel times, roofline of the dominant kernel, parity
    and a CPU baseline on a small sample."""
    B, NP, res = args.dense_pairs, args.points, args.res
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=dev)
    pr = synth.pair_2d(seeds, NP, device=dev, chunk_bytes=1 << 30, scene="dense")
    both = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
    T_init_cm = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
    idx = torch.arange(B, dtype=torch.int32, device=dev)
    idx_src = idx + B
    ms = N.MapSet(res, [0, 0, 0], size_m, n_maps=2 * B, max_cells=4096)
    ms.profiling(True)
    T16 = T_init_cm.clone()
    results = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream()

    def step():
        ms.build(both, range_limit=rng_lim, stream=st)
        T16.copy_(T_init_cm)
        binding.match_batch_device(ms, idx, ms, idx_src, T16, results, B, stream=st)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    kb, km = [], []
    t0 = time.perf_counter()
    n_steps = 5
    for _ in range(n_steps):
        step()
        torch.cuda.synchronize()                       # (serial: the events of the library bracket each kernel alone)
        kb.append(ms.last_kernel_ms(0)); km.append(ms.last_kernel_ms(1))
    elapsed = time.perf_counter() - t0
    r = results.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
    T_out = T16.cpu().numpy().reshape(B, 4, 4).transpose(0, 2, 1)
    cells = float(r["n_target"].astype(np.int64).sum() + r["n_source"].astype(np.int64).sum())
    build_ms, match_ms = float(np.median(kb)), float(np.median(km))
    build_bytes = 2 * B * 12.0 * NP + 80.0 * cells
    gflop = (130.0 * float(r["pair_terms_g"].sum()) + 610.0 * float(r["pair_terms_h"].sum())) / 1e9
    if build_ms >= match_ms:
        roof = {"kernel": "ndt_build_kernel", "bound": "hbm", "achieved": build_bytes / build_ms / 1e6, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": build_bytes / build_ms / 1e6 / HBM_PEAK_GBS, "traffic": None}
    else:
        roof = {"kernel": "ndt_match_kernel", "bound": "fp64_valu", "achieved": gflop / match_ms, "peak": 78.6, "unit": "TFLOP/s",
                "frac": gflop / match_ms / 78.6, "traffic": None}
    # the same steps through the registrar (ONE ndtgpu_register_batch_device call per step, three internal map sets / streams):
    # the builds and the first registrations of step k + 1 fill the CUs that the long registrations of step k have left
    ms.profiling(False)
    D = max(1, min(8, int(args.buffers)))
    reg = N.Registrar(res, [0, 0, 0], size_m, pairs_per_batch=B, depth=D, max_cells=4096)
    outs = [(T_init_cm.clone(), torch.zeros((B, 64), dtype=torch.uint8, device=dev), [0]) for _ in range(D)]
    n_pipe = 32            # (12 until round 6: fill and drain of the eight-deep ring were a fifth of that region)

    def pstep(k):
        T_k, r_k, tk = outs[k % D]
        if tk[0]:
            reg.wait_stream(st, ticket=tk[0])
        T_k.copy_(T_init_cm)
        tk[0] = reg.submit(both[:B], both[B:], T_k, r_k, range_limit=rng_lim, stream=st)
    for k in range(D):
        pstep(k)
    reg.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n_pipe):
        pstep(k)
    reg.sync(); torch.cuda.synchronize()
    elapsed_pipe = time.perf_counter() - t0
    same = bool(torch.equal(outs[(n_pipe - 1) % D][0], T16) and torch.equal(outs[(n_pipe - 1) % D][1][:, :32], results[:, :32]))
    reg_info = reg.info()
    reg.close()
    out = {"value": B * n_pipe / elapsed_pipe, "unit": "registrations/s", "pairs": B, "steps": n_pipe, "registrar": reg_info,
           "ms_per_step": 1e3 * elapsed_pipe / n_pipe, "value_serial": B * n_steps / elapsed,
           "ms_per_step_serial": 1e3 * elapsed / n_steps, "pipelined_equals_serial_bits": same,
           "workload": "%d pairs x %d pts, %.2f m cells, synth scene 'dense' (the hall of the headline scene + 3000 posts of 2-4 cm), "
                       "fuser preset; value: ONE ndtgpu_register_batch_device call per step through a default registrar (eight map sets, "
                       "the split of the chip measured on the first sub-batch: `registrar`); *_serial and kernel_ms: the two launches alone on one stream" % (B, NP, res),
           "mean_cells_per_map": cells / (2 * B), "kernel_ms": {"ndt_build_kernel": build_ms, "ndt_match_kernel": match_ms},
           "build_hbm_frac": build_bytes / build_ms / 1e6 / HBM_PEAK_GBS, "match_fp64_frac": gflop / match_ms / 78.6,
           "pair_terms_per_registration": float((r["pair_terms_g"].sum() + r["pair_terms_h"].sum()) / B),
           "mean_iterations": float(r["iterations"].mean()), "converged_frac": float(r["converged"].mean()), "roofline": roof}
    if with_cpu:
        S = min(args.dense_cpu_sample, B)
        one, omp, To = cpu_baseline_c(pr["fixed"][:S].cpu().numpy(), pr["moving"][:S].cpu().numpy(), pr["T_init"][:S].cpu().numpy(),
                                      res, size_m, rng_lim, 1e-6, 2, 30, reps=3, tag="dense")
        max_dt = max_dr = 0.0
        for b in range(S):
            max_dt = max(max_dt, float(np.linalg.norm(T_out[b][:3, 3] - To[b][:3, 3])))
            max_dr = max(max_dr, float(2 * np.arcsin(min(1.0, np.linalg.norm(T_out[b][:3, :3] - To[b][:3, :3]) / (2 * np.sqrt(2))))))
        out["cpu_baseline"] = {"value": one["registrations_per_s"], "unit": "registrations/s", "cores": 1, "kind": "port",
                               "sample": "first %d pairs of the dense batch, oracle/cpu_baseline.c pinned to one core, median of %d passes "
                                         "(%.2f s per pass)" % (S, one["reps"], one["median_pass_s"]),
                               "all_cores": {"value": omp["registrations_per_s"], "threads": omp["threads"]}}
        out["parity"] = {"pairs_checked": S, "max_dt_m": max_dt, "max_drot_rad": max_dr, "tolerance": "1e-4 m / 1e-4 rad",
                         "ok": bool(max_dt <= 1e-4 and max_dr <= 1e-4)}
        out["speedup_vs_cpu_1thread"] = out["value"] / one["registrations_per_s"]
    return out


def _masked_stream(torch, dev, first_cu, n_cus, n_cu_dev):
    """A HIP stream whose kernels only run on CUs [first_cu, first_cu + n_cus) (hipExtStreamCreateWithCUMask)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    words = (n_cu_dev + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in range(first_cu, first_cu + n_cus):
        mask[cu // 32] |= 1 << (cu % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), C.c_uint32(words), mask)
    if rc != 0 or not h.value:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(h.value, device=dev)


def _timed(torch, fn, reps=5, warm=2):
    """median wall time [ms] of fn() bracketed by device synchronisation"""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - c0))
    return float(np.median(ts))


def config5(args, torch, N, binding, synth, dev):
    """BASELINE configs[4]: 3D mode, 200 k-point Velodyne-style clouds, 0.25 m voxels, 6-DoF.  One pair through the
    host-synchronous call (latency: the reference's call shape) and a batch of 64 sweeps = 32 pairs resident in HBM
    (throughput).  Rooflines: the build against HBM (12 N + 80 M bytes per sweep), the matcher against the fp64 peak."""
    res, size, rng, cap = 0.25, [100.0, 100.0, 10.0], 70.0, 120000
    B = 32
    pr = synth.pair_3d(torch.arange(1, B + 1, device=dev), device=dev)
    sweeps = torch.cat([pr["fixed"], pr["moving"]]).contiguous()                  # [64, 200000, 3]
    NP = int(sweeps.shape[1])
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=2 * B, max_cells=cap)
    ms.profiling(True)
    st = torch.cuda.current_stream()
    Ti = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)
    T16 = Ti.clone()
    results = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
    ti = torch.arange(B, dtype=torch.int32, device=dev)
    si = ti + B
    build_ms = _timed(torch, lambda: ms.build(sweeps, range_limit=rng, stream=st))
    cells = ms.num_cells_all().astype(np.float64)

    def match():
        T16.copy_(Ti)
        binding.match_batch_device(ms, ti, ms, si, T16, results, B, stream=st)
    match_ms = _timed(torch, match)
    r = results.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
    step_ms = _timed(torch, lambda: (ms.build(sweeps, range_limit=rng, stream=st), match()))
    build_bytes = 2 * B * 12.0 * NP + 80.0 * cells.sum()
    gflop = (130.0 * float(r["pair_terms_g"].sum()) + 610.0 * float(r["pair_terms_h"].sum())) / 1e9
    # one pair, host-synchronous (cooperative launches: the registration is spread over the chip)
    one = N.MapSet(res, [0, 0, 0], size, n_maps=2, max_cells=cap)
    two = torch.stack([pr["fixed"][0], pr["moving"][0]]).contiguous()
    T0 = pr["T_init"][0].cpu().numpy()
    pair_ms = _timed(torch, lambda: (one.build(two, range_limit=rng, stream=st), N.match_d2d(one, 0, one, 1, T0)))
    out = {"metric": "NDT scan-pair registrations/sec (100k pts, 0.5 m cells)", "value": B / (step_ms * 1e-3), "unit": "registrations/s",
           "n_gpus": 1, "steps": 5, "warmup": 2, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "configs[4]: 3D mode, %d sweeps of %d points (64 rings), 0.25 m voxels, grid 100x100x10 m, range 70 m, "
                                  "6-DoF D2D, n_neighbours 2; one step = build of the 64 sweeps + match of the 32 pairs (serial, one stream)" % (2 * B, NP),
                      "mean_cells_per_map": float(cells.mean())},
           "roofline": {"kernel": "ndt_build_kernel (MODE 1 accumulate + MODE 2/3 finalise, 64 sweeps)", "bound": "hbm",
                        "achieved": build_bytes / build_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": build_bytes / build_ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                        "note": "algorithmic bytes 12 N + 80 M per sweep / wall time of the build call (four launches: accumulate, "
                                "moments -> Gaussians, ranking, placement); instruction-issue bound: a ring crosses a cell "
                                "every ~6 points, ~1 flush record per 6 points, 16 % of the lanes of a point step push one "
                                "(DESIGN.md 4.1b, 7)"},
           "kernels": {"build_64_sweeps": {"ms": build_ms, "points_per_s": 2 * B * NP / (build_ms * 1e-3), "algorithmic_bytes": build_bytes},
                       "match_32_pairs": {"ms": match_ms, "fp64_gflop": gflop, "fp64_tflops": gflop / match_ms,
                                          "frac_of_fp64_peak": gflop / match_ms / 78.6, "mean_iterations": float(r["iterations"].mean()),
                                          "converged_frac": float(r["converged"].mean()),
                                          "note": "task-pool matcher (a batch that cannot fill the chip with one workgroup per registration): one "
                                                  "asynchronous launch, any workgroup takes any (registration, evaluation, two 256-cell chunks) task; "
                                                  "NDTGPU_POOL=0: static teams at a grid barrier, 10.7 ms; NDTGPU_DEVICE_COOP=0: persistent kernel, one "
                                                  "CU per registration, 65 ms"}},
           "single_pair": {"ms_build_x2_plus_match": pair_ms, "note": "host-synchronous ndtgpu_mapset_build + ndtgpu_match_d2d (grid-barrier "
                                                                         "kernel, a workgroup per 256-cell chunk)"}}
    print(json.dumps(out))


def config_fuse(args, torch, N, binding, synth, dev):
    """The node-map path of SURVEY 8(f): ndtgpu_mapset_add_cloud (= NDTMap::addPointCloud + computeNDTCells: ray tracing,
    accumulate, finalise) of 256 node maps with one 100 k-point cloud each, then the per-link work of the offline
    refinement on 200 fused node maps (16 scans each): matcher, NDTMatcherD2D::covariance and overlapNDTOccupancyScore of
    all 19 900 links.  Algorithmic bytes per add_cloud: 12 N (points) + 12 slots (occupancy + evidence) + 2 x 80 M (cells)."""
    res, size, rng = args.res, [100.0, 100.0, 1.0], 30.0
    NP, Bm = args.points, 256
    poses = torch.zeros((Bm, 3), dtype=torch.float64)
    poses[:, 0] = torch.linspace(-1.5, 1.5, Bm, dtype=torch.float64)
    scans = synth.scan_2d(torch.full((Bm,), 77, dtype=torch.int64, device=dev), poses.to(dev), NP, chunk_bytes=2 << 30).contiguous()
    ms = N.MapSet(res, [0, 0, 0], size, n_maps=Bm, max_cells=4096)
    ms.enable_occupancy()
    origins = np.concatenate([poses[:, :2].numpy(), np.zeros((Bm, 1))], axis=1)
    st = torch.cuda.current_stream()

    def add():
        ms.clear()
        ms.add_cloud(scans, origins, stream=st, maxz=100.0, sensor_noise=0.1)
    add_ms = _timed(torch, add)
    clear_ms = _timed(torch, lambda: ms.clear())
    cells = ms.num_cells_all().astype(np.float64)
    slots = int(size[0] / res) * int(size[1] / res) * max(1, int(size[2] / res))
    add_bytes = Bm * (12.0 * NP + 12.0 * slots) + 2 * 80.0 * cells.sum()
    # 200 fused node maps of 16 scans (20 k points each), all-pairs links
    nn, ns, npts = 200, 16, 20000
    nodes = N.MapSet(res, [0, 0, 0], size, n_maps=nn, max_cells=4096)
    nodes.enable_occupancy()
    t = torch.linspace(0.0, 2.0 * np.pi, nn + 1, dtype=torch.float64)[:-1]
    node_pose = torch.stack([1.6 * torch.sin(t), 1.2 * torch.sin(2.0 * t + 0.3), 0.35 * torch.sin(3.0 * t)], dim=1)
    node_T = synth.pose2d_to_T(node_pose).numpy()
    c0 = time.perf_counter()
    for k in range(ns):
        p = node_pose.clone()
        p[:, 0] += 0.01 * k * torch.cos(node_pose[:, 2]); p[:, 1] += 0.01 * k * torch.sin(node_pose[:, 2])
        sc = synth.scan_2d(torch.full((nn,), 321, dtype=torch.int64, device=dev), p.to(dev), npts).contiguous()
        # the cloud in the node frame: the scan frame is 1 cm x k ahead of the node frame
        sc[:, :, 0] += 0.01 * k
        org = np.tile(np.array([[0.01 * k, 0.0, 0.0]]), (nn, 1))
        nodes.add_cloud(sc, org, stream=st, **(dict(maxz=100.0, sensor_noise=0.1) if k == 0 else dict(maxz=25.0, sensor_noise=0.06)))
    torch.cuda.synchronize()
    fuse_total_ms = 1e3 * (time.perf_counter() - c0)
    iu = np.triu_indices(nn, 1)
    ti, si = iu[0].astype(np.uint32), iu[1].astype(np.uint32)
    T0 = np.einsum("eij,ejk->eik", np.linalg.inv(node_T[ti]), node_T[si])
    n_links = len(ti)
    holder = {}

    def m():
        holder["T"], holder["r"] = N.match_batch(nodes, ti, nodes, si, T0, delta_score=1e-3)
    match_ms = _timed(torch, m, reps=3, warm=1)
    cov_ms = _timed(torch, lambda: binding.covariance(nodes, ti, nodes, si, holder["T"]), reps=3, warm=1)
    ovl_ms = _timed(torch, lambda: binding.overlap_score(nodes, ti, nodes, si, holder["T"]), reps=3, warm=1)
    r = holder["r"]
    mcells = nodes.num_cells_all().astype(np.float64)
    gflop = (130.0 * float(r["pair_terms_g"].sum()) + 610.0 * float(r["pair_terms_h"].sum())) / 1e9
    cov_bytes = n_links * 80.0 * 2 * mcells.mean()          # both cell maps of a link, once (they then live in L2)
    ovl_bytes = n_links * 4.0 * slots * 2                   # the occupancy arrays of both maps
    out = {"metric": "NDT scan-pair registrations/sec (100k pts, 0.5 m cells)", "value": n_links / (match_ms * 1e-3), "unit": "registrations/s",
           "n_gpus": 1, "steps": 3, "warmup": 1, "ms_per_step": match_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "node-map path (SURVEY 8f): add_cloud of %d node maps x %d points; %d links of %d fused node maps (%d scans x %d "
                                  "points each): matcher (edge preset), covariance, occupancy overlap; host arrays in and out" % (Bm, NP, n_links, nn, ns, npts),
                      "mean_cells_fused_map": float(mcells.mean())},
           "roofline": {"kernel": "ndtgpu_mapset_add_cloud = ndt_raytrace_kernel + ndt_build_kernel<.,1> + ndt_fuse_finalize_kernel", "bound": "hbm",
                        "achieved": add_bytes / add_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": add_bytes / add_ms / 1e6 / HBM_PEAK_GBS,
                        "traffic": None,
                        "note": "algorithmic bytes (12 N + 12 slots) per map + 2 x 80 M / wall time of the call (three launches, includes the "
                                "clear of the previous content: %.3f ms); the ray walk is VALU bound (~30 samples per beam, 3 exact cell "
                                "indices per sample), profiles/r03_fuse_*" % clear_ms},
           "kernels": {"add_cloud_256_maps": {"ms": add_ms, "us_per_scan": 1e3 * add_ms / Bm, "algorithmic_bytes": add_bytes},
                       "fuse_200_nodes_16_scans": {"ms_total_incl_synthesis": fuse_total_ms},
                       "match_links": {"ms": match_ms, "us_per_link": 1e3 * match_ms / n_links, "fp64_gflop": gflop,
                                       "frac_of_fp64_peak": gflop / match_ms / 78.6, "converged_frac": float(r["converged"].mean()),
                                       "mean_iterations": float(r["iterations"].mean())},
                       "covariance_links": {"ms": cov_ms, "us_per_link": 1e3 * cov_ms / n_links, "algorithmic_bytes": cov_bytes,
                                            "GBps": cov_bytes / cov_ms / 1e6},
                       "overlap_links": {"ms": ovl_ms, "us_per_link": 1e3 * ovl_ms / n_links, "algorithmic_bytes": ovl_bytes,
                                         "GBps": ovl_bytes / ovl_ms / 1e6, "frac_of_hbm_peak": ovl_bytes / ovl_ms / 1e6 / HBM_PEAK_GBS}}}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=1024, help="scan pairs per GPU per step")
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--res", type=float, default=0.5)
    ap.add_argument("--cpu-sample", type=int, default=128, help="pairs timed on the CPU oracle (0 = skip): 6 passes of ~2 s")
    ap.add_argument("--config", type=str, default="3", choices=["3", "4", "5", "fuse"],
                    help="3: BASELINE configs[2], the batch the metric is quoted on (default); 4: configs[3], the graph replay harness; "
                         "5: configs[4], 3D mode; fuse: the node-map path (add_cloud, link covariance and overlap)")
    ap.add_argument("--nodes", type=int, default=5000, help="--config 4: node maps (5000 = the full config)")
    ap.add_argument("--gated", action="store_true", help="--config 4: (default since round 4) time the candidate edges within --gate-dist")
    ap.add_argument("--all-pairs", action="store_true", help="--config 4: time all pairs of computeAllPossibleLinks instead of the gated ones")
    ap.add_argument("--no-all-pairs", action="store_true", help="--config 4: skip the untimed all-pairs extra")
    ap.add_argument("--gate-dist", type=float, default=6.0, help="--config 4: candidate gate on the odometry distance of two nodes [m]")
    ap.add_argument("--scans-per-node", type=int, default=10, help="--config 4: scans fused into every node map")
    ap.add_argument("--node-points", type=int, default=20000, help="--config 4: points per scan")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dense-pairs", type=int, default=384, help="pairs of the dense-scene leg (0 = skip it)")
    ap.add_argument("--dense-cpu-sample", type=int, default=16, help="pairs of the dense-scene leg timed on the CPU")
    ap.add_argument("--buffers", type=int, default=8, help="pipeline depth (the registrar's internal map sets; at most 8)")
    ap.add_argument("--cu-split", type=int, default=0, help="CUs given to the build streams (hipExtStreamCreateWithCUMask), the "
                    "matcher streams get the rest; 0: every stream sees the whole chip")
    ap.add_argument("--registrar", type=str, default="", help="fields of ndtgpu_registrar_params for the timed registrar, e.g. "
                    "'matcher_groups=128,recalibrate_pct=-1' (experiments; the default registrar is what the headline is quoted on)")
    ap.add_argument("--sub-batch", type=int, default=0, help="pairs per internal sub-batch of the registrar (0: --pairs, one sub-batch per call)")
    ap.add_argument("--legacy-pipeline", action="store_true",
                    help="the round-4 form: bench.py itself drives mapset pairs, streams and events (ndtgpu_mapset_build + "
                         "ndtgpu_match_batch_device per step) instead of ONE ndtgpu_register_batch_device call per step")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one stream, one mapset pair: every step waits for the previous one (default: --buffers mapset pairs, the "
                         "grid builds of step k+1 run on the CUs the matcher of step k has already left)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ndt_feature_graph_amd as N
    from ndt_feature_graph_amd import binding, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available() or N.device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from ndt_feature_graph_amd import distributed as _D
    use_dist = world > 1 or _D.collectives_forced()     # NDTGPU_FORCE_COLLECTIVES=1: the RCCL path on ONE rank (tests/test_gpu_rccl.py)
    args.use_dist = use_dist
    if use_dist:
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, NP, res = args.pairs, args.points, args.res
    size_m = [100.0, 100.0, 1.0]           # gustav_laser_tf.launch:16-18
    rng_lim = 30.0                          # sensor_range, launch:22
    if args.config == "4":
        return config4(args, torch, dist, N, binding, synth, rank, world, dev, size_m, rng_lim)
    if args.config == "5":
        return config5(args, torch, N, binding, synth, dev)
    if args.config == "fuse":
        return config_fuse(args, torch, N, binding, synth, dev)

    # ---- synthetic batch, generated on the GPU, resident in HBM before the timed region -----
    seeds = torch.arange(1 + rank * B, 1 + (rank + 1) * B, dtype=torch.int64, device=dev)
    pr = synth.pair_2d(seeds, NP, device=dev, chunk_bytes=2 << 30)
    # the 2B scans of a step sit in ONE tensor (fixed scans first): one build launch makes all 2B cell maps of one
    # map set; registration i matches map i (target) against map B + i (source)
    both = torch.cat([pr["fixed"], pr["moving"]]).contiguous()
    pr["fixed"] = pr["moving"] = None
    fixed, moving = both[:B], both[B:]
    T_init_cm = pr["T_init"].transpose(1, 2).contiguous().reshape(B, 16)     # column-major Affine3d
    idx = torch.arange(B, dtype=torch.int32, device=dev)
    idx_src = idx + B

    # The timed step is ONE C-ABI call: ndtgpu_register_batch_device (scans in HBM -> poses).  The registrar owns `n_buf`
    # internal map sets and streams; the builds of step k+1 are released when the builds of step k are done, i.e. while the
    # matcher of step k runs: its workgroups leave their CUs as soon as no registration is left to start
    # (csrc/ndt_match.hip), so the next step's builds fill the CUs that the few long registrations of this step do not
    # occupy.  Every step still does all of its work; results are identical to the serial run.  (Until round 4 this
    # choreography lived here, in Python: --legacy-pipeline keeps it for A/B.)
    n_buf = 1 if args.no_pipeline else args.buffers
    n_cu_dev = torch.cuda.get_device_properties(dev).multi_processor_count
    legacy = args.legacy_pipeline or args.cu_split > 0
    if legacy:
        n_buf = min(n_buf, 4)                  # (the round-4 choreography: a whole map set per buffer)
    n_buf = max(1, min(n_buf, 8))              # (the registrar's ring holds at most eight map sets)
    if args.cu_split > 0 and n_buf > 1:
        os.environ["NDTGPU_MATCH_GROUPS"] = str(n_cu_dev - args.cu_split)

    class Buf:
        pass
    bufs = []
    reg_fields = {kv.split("=")[0].strip(): int(kv.split("=")[1]) for kv in args.registrar.split(",") if "=" in kv}
    reg = None if legacy else N.Registrar(res, [0, 0, 0], size_m, pairs_per_batch=(args.sub_batch or B), depth=n_buf, max_cells=4096, **reg_fields)
    main_stream = torch.cuda.current_stream()
    if os.environ.get("BENCH_MAIN_SIDE"):      # (experiment: the caller's stream is not the null stream)
        main_stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(main_stream)
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None
    for k in range(n_buf):
        b = Buf()
        b.maps = N.MapSet(res, [0, 0, 0], size_m, n_maps=2 * B, max_cells=4096) if legacy else reg.mapset(k)
        b.T16 = T_init_cm.clone()
        b.results = torch.zeros((B, 64), dtype=torch.uint8, device=dev)
        b.stream = torch.cuda.Stream(device=dev) if legacy else None      # (the registrar brings its own streams)
        b.mstream = b.stream
        if args.cu_split > 0 and n_buf > 1:
            # builds on the first `cu_split` CUs, matchers on the rest: the matcher's one-per-CU workgroups never wait for a CU
            # to drain its small build workgroups, and the builds never lose CUs to a matcher's tail
            b.stream = _masked_stream(torch, dev, 0, args.cu_split, n_cu_dev)
            b.mstream = _masked_stream(torch, dev, args.cu_split, n_cu_dev - args.cu_split, n_cu_dev)
        b.match_done = None
        b.ticket = 0
        b.gathered = None
        b.gathered_ev = None
        if use_dist:
            b.gathered = [torch.empty((world * B, 16), dtype=torch.float64, device=dev),
                          torch.empty((world * B, 64), dtype=torch.uint8, device=dev)]
        bufs.append(b)
    state = {"k": 0, "match_started": None}

    # (experiment: BENCH_ITR_MAX=1 leaves the matcher next to nothing to do -- what the build side alone delivers at the split)
    exp_params = {"itr_max": int(os.environ["BENCH_ITR_MAX"])} if os.environ.get("BENCH_ITR_MAX") else {}

    def step_registrar(ev=None):
        b = bufs[state["k"] % n_buf]
        state["k"] += 1
        if b.ticket:                                   # the matcher of step k - n_buf wrote b.T16 / b.results
            reg.wait_stream(main_stream, ticket=b.ticket)
        if b.gathered_ev is not None:
            main_stream.wait_event(b.gathered_ev)      # ... and its all-gather read them
        b.T16.copy_(T_init_cm)
        b.ticket = reg.submit(both[:B], both[B:], b.T16, b.results, range_limit=rng_lim, stream=main_stream, **exp_params)
        if use_dist:   # final gather of the edge transforms (the only collective on the path)
            reg.wait_stream(comm_stream, ticket=b.ticket)
            with torch.cuda.stream(comm_stream):
                dist.all_gather_into_tensor(b.gathered[0], b.T16)
                dist.all_gather_into_tensor(b.gathered[1], b.results)
                b.gathered_ev = torch.cuda.Event()
                b.gathered_ev.record(comm_stream)

    def step_legacy(ev=None):
        b = bufs[state["k"] % n_buf]
        state["k"] += 1
        st, mst = b.stream, b.mstream
        with torch.cuda.stream(st):
            if n_buf > 1 and state["match_started"] is not None:
                st.wait_event(state["match_started"])
            if mst is not st and b.match_done is not None:
                st.wait_event(b.match_done)            # this buffer's maps are still being matched (step k - n_buf)
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if ev is not None else None
            if marks: marks[0].record(st)
            b.maps.build(both, range_limit=rng_lim, stream=st)
            if marks: marks[1].record(st); marks[2].record(st)
            started = torch.cuda.Event()
            started.record(st)
            state["match_started"] = started
        with torch.cuda.stream(mst):
            if mst is not st:
                mst.wait_event(started)
            b.T16.copy_(T_init_cm)
            if marks:
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[4].record(mst)
            binding.match_batch_device(b.maps, idx, b.maps, idx_src, b.T16, b.results, B, stream=mst)
            if marks:
                marks[3].record(mst)
                ev.append(marks)
            if use_dist:   # final gather of the edge transforms (the only collective on the path)
                dist.all_gather_into_tensor(b.gathered[0], b.T16)
                dist.all_gather_into_tensor(b.gathered[1], b.results)
            if mst is not st:
                b.match_done = torch.cuda.Event()
                b.match_done.record(mst)

    step = step_legacy if legacy else step_registrar

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    # isolated kernel durations (a serial step on a plain map set: build launch, then matcher launch, alone on the chip;
    # events inside the library), outside the timed region
    # (a map set of its own: the registrar's sets keep their matcher launches on part of the chip)
    iso = bufs[0].maps if legacy else N.MapSet(res, [0, 0, 0], size_m, n_maps=2 * B, max_cells=4096)
    iso.profiling(True)
    iso_T16 = T_init_cm.clone()
    iso_b, iso_m = [], []
    for _ in range(6):
        iso.build(both, range_limit=rng_lim, stream=main_stream)
        iso_T16.copy_(T_init_cm)
        binding.match_batch_device(iso, idx, iso, idx_src, iso_T16, bufs[0].results, B, stream=main_stream)
        barrier()
        iso_b.append(iso.last_kernel_ms(0)); iso_m.append(iso.last_kernel_ms(1))
    iso_build_ms = float(np.median(iso_b[1:]))     # one launch: 2B scans (median of five warm serial steps)
    iso_match_ms = float(np.median(iso_m[1:]))
    iso.profiling(False)
    if not legacy:
        iso.close()
    for _ in range(n_buf):     # one serial step per buffer: code objects loaded, every buffer's pages touched
        step()
        if reg is not None:
            reg.sync()
        barrier()
        state["match_started"] = None
    state["k"] = 0

    for _ in range(args.warmup):
        step()
    if reg is not None:
        reg.sync()
        reg.profiling(True)    # HIP events on the registrar's internal streams bracket every launch of the timed region
    barrier()
    marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(marks)          # (legacy pipeline: HIP events on the launch stream bracket every kernel of the timed region)
    if reg is not None:
        reg.sync()           # the host waits for the registrar's streams (the caller's stream was never made to wait)
    barrier()
    elapsed = time.perf_counter() - t0
    if reg is not None:
        build_ms_ev, match_ms_ev, n_prof = reg.kernel_ms()
        assert n_prof >= args.steps
        # the stream-fed matcher: ONE instance serves batch after batch, there is no launch per step -- what the registrar reports as
        # its matcher side is then the queue's own stamp of a batch: publication until its last registration finished (a latency, not
        # a share of the chip's time)
        stream_fed = reg.info()["matcher_form"] == binding.MATCHER_STREAM_FED
        batch_latency_ms = match_ms_ev if stream_fed else None
        if stream_fed:
            match_ms_ev = None
        k_build, k_match = [build_ms_ev], [match_ms_ev if match_ms_ev is not None else 1e3 * elapsed / args.steps]
        reg.profiling(False)
    else:
        k_build = [m[0].elapsed_time(m[1]) for m in marks]
        k_match = [m[4].elapsed_time(m[3]) for m in marks]
    # ---- labelled extra, outside the timed region: the same steps over a region long enough for the ring's fill and drain not to
    # matter (a burst pays one pipeline latency, ~3.6 ms, whatever its length: a tenth of 20 steps, a fortieth of 100)
    steady = None
    if reg is not None and args.steps < 100 and not args.no_pipeline:
        barrier()
        ts0 = time.perf_counter()
        for _ in range(100):
            step()
        reg.sync()
        barrier()
        steady_s = time.perf_counter() - ts0
        if use_dist:
            tm2 = torch.tensor([steady_s], dtype=torch.float64, device=dev)
            dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
            steady_s = float(tm2.item())
        steady = {"value": world * B * 100 / steady_s, "unit": "registrations/s", "steps": 100, "ms_per_step": 1e3 * steady_s / 100,
                  "note": "the same call, 100 steps between the same barriers, after the timed region: not the headline value"}
    last = bufs[(state["k"] - 1) % n_buf]
    T16, results = last.T16, last.results
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed

    # ---- bookkeeping outside the timed region -------------------------------------------------
    res_np = results.cpu().numpy().view(binding.RESULT_DTYPE).reshape(B)
    T_out = T16.cpu().numpy().reshape(B, 4, 4).transpose(0, 2, 1)
    m_t, m_s = res_np["n_target"].astype(np.int64), res_np["n_source"].astype(np.int64)
    build_ms = float(np.mean(k_build))                                 # per launch (2B scans)
    match_ms = float(np.mean(k_match))
    # algorithmic bytes: build 12*N + 80*M per scan; match 80*(M_src+M_tgt) per pair
    build_bytes = 2 * B * 12.0 * NP + 80.0 * float(m_t.sum() + m_s.sum())           # per launch
    match_bytes = 80.0 * float(m_t.sum() + m_s.sum())
    kern = {
        "ndt_build_kernel": {"ms_per_launch": build_ms, "ms_isolated": iso_build_ms, "launches_per_step": 1,
                             "scans_per_launch": 2 * B,
                             "algorithmic_bytes": build_bytes, "GBps": build_bytes / build_ms / 1e6,
                             "GBps_isolated": build_bytes / iso_build_ms / 1e6},
        "ndt_match_kernel": {"ms_per_launch": match_ms, "ms_isolated": iso_match_ms, "launches_per_step": 1,
                             "algorithmic_bytes": match_bytes,
                             "GBps": match_bytes / match_ms / 1e6,
                             # k-bar, E and the matcher's fp64 work (SURVEY.md 8d asks for them as measured outputs)
                             "pair_terms_per_launch": int(res_np["pair_terms_g"].sum() + res_np["pair_terms_h"].sum()),
                             "mean_neighbours_kbar": float((res_np["pair_terms_g"].sum() + res_np["pair_terms_h"].sum())
                                                           / max(1, int((res_np["fevals"].astype(np.int64) * m_s).sum()))),
                             "fp64_gflop_per_launch": (130.0 * float(res_np["pair_terms_g"].sum())
                                                       + 610.0 * float(res_np["pair_terms_h"].sum())) / 1e9,
                             "mean_iterations": float(res_np["iterations"].mean()),
                             "mean_fevals": float(res_np["fevals"].mean()),
                             "converged_frac": float(res_np["converged"].mean())},
    }
    mk = kern["ndt_match_kernel"]
    mk["fp64_tflops"] = mk["fp64_gflop_per_launch"] / match_ms          # GFLOP / ms = TFLOP/s
    mk["fp64_note"] = ("pair-term flops: 130 per gradient term, 610 per Hessian term -- the per-term figures of the formulation in "
                       "csrc/ndt_match.hip, the unit of every round's line (DESIGN.md 4.2); the shipped loops EXECUTE 145 / 394 fp64 "
                       "flops per term (ISA count, fma = 2: since library 0.5.8 the Hessian term no longer issues its 42 "
                       "multiplications by the structural zeros of e_k x v, since 0.5.9 its rotational blocks are d_i^T B d_k with "
                       "d_k = j_k - r_k), i.e. %.2f GFLOP per launch; "
                       "MI355X fp64 vector peak 78.6 TFLOP/s (AMD datasheet) -> frac %.4f"
                       % ((145.0 * float(res_np["pair_terms_g"].sum()) + 394.0 * float(res_np["pair_terms_h"].sum())) / 1e9,
                          mk["fp64_tflops"] / 78.6))
    mk["fp64_gflop_executed_per_launch"] = (145.0 * float(res_np["pair_terms_g"].sum()) + 394.0 * float(res_np["pair_terms_h"].sum())) / 1e9
    # What the REFERENCE's loop needs: per registration one evaluation with the Hessian per Newton iteration
    # (fusion.h:856) and gradient-only evaluations for everything else it runs -- every line-search trial, the re-evaluation of
    # the Newton pose that opens a line search (fusion.h:444; the kernel reuses the Newton sums and neither runs nor counts it),
    # the score at the returned pose (fusion.h:1085): `fevals` + `iterations` evaluations, `iterations` of them with the Hessian.
    # Terms per evaluation of a registration: its own mean.  Hessians the kernel evaluates on speculation (trials from the second
    # on, the first trial while first trials are accepted) are NOT in this figure.
    ev_terms = (res_np["pair_terms_g"] + res_np["pair_terms_h"]).astype(np.float64) / np.maximum(1, res_np["fevals"]).astype(np.float64)
    mk["fp64_gflop_needed_per_launch"] = float((ev_terms * (610.0 * res_np["iterations"] + 130.0 * res_np["fevals"])).sum()) / 1e9
    dominant = "ndt_build_kernel" if iso_build_ms >= iso_match_ms else "ndt_match_kernel"   # by time alone on the chip
    dk = kern[dominant]
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command
    # (tools/collect_profiles.sh -> profiles/rNN_pmc_traffic.json; FETCH_SIZE x2 on gfx950 + WRITE_SIZE)
    traffic = None
    traffic_note = "no PMC summary for this library version under profiles/ (tools/collect_profiles.sh): traffic = null"
    lib_version = binding.lib().ndtgpu_version().decode()
    try:
        if B == 1024 and NP == 100000:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            # a summary taken with another binary says nothing about this one: refuse it
            if pmc.get("lib_version") == lib_version:
                for k in kern:
                    kern[k]["pmc_hbm_bytes_per_launch"] = pmc["kernels"][k]["hbm_bytes_per_launch"]
                traffic = pmc["kernels"][dominant]["hbm_bytes_per_launch"]
                traffic_note = ("traffic = HBM bytes per launch from the rocprofv3 PMC passes of this library version committed as "
                                "profiles/%s" % PMC_FILE)
    except Exception:
        traffic = None
    note = ("achieved / frac = algorithmic work per launch / HIP-event duration of the launch ALONE on the chip (events on the "
            "launch stream, the median of five warm serial steps of this run before the timed region; agrees with profiles/%s_bench_kernel_stats_serial.csv); "
            "*_timed_region = the same work / the HIP-event duration of the launch inside the timed region, where it shares the chip "
            "with the neighbouring steps' kernels (events on the registrar's internal streams); " % ROUND_TAG + traffic_note)
    if dominant == "ndt_build_kernel":       # streaming pass over the points: HBM roof
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": dk["algorithmic_bytes"] / dk["ms_isolated"] / 1e6, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": dk["algorithmic_bytes"] / dk["ms_isolated"] / 1e6 / HBM_PEAK_GBS, "traffic": traffic,
                    "achieved_timed_region": dk["GBps"], "frac_timed_region": dk["GBps"] / HBM_PEAK_GBS, "note": note}
    else:
        # the matcher re-reads two cell maps that live in L2 ~1000 times: its roof is fp64 arithmetic,
        # 78.6 TFLOP/s on MI355X for matrix and vector fp64 alike.  Flops = the kernel's own pair-term counters x
        # 130 (gradient term) / 610 (Hessian term), DESIGN.md 4.2.  kernels.ndt_match_kernel.GBps is the HBM view.
        roofline = {"kernel": dominant, "bound": "fp64_valu", "achieved": mk["fp64_gflop_per_launch"] / mk["ms_isolated"], "peak": 78.6,
                    "unit": "TFLOP/s", "frac": mk["fp64_gflop_per_launch"] / mk["ms_isolated"] / 78.6, "traffic": traffic,
                    "achieved_timed_region": mk["fp64_tflops"], "frac_timed_region": mk["fp64_tflops"] / 78.6,
                    "frac_step": mk["fp64_gflop_per_launch"] / ms_per_step / 78.6,
                    # the same launch priced by the fp64 flops its loops EXECUTE (ISA count; see kernels.ndt_match_kernel.fp64_note)
                    "frac_executed_flops": mk["fp64_gflop_executed_per_launch"] / mk["ms_isolated"] / 78.6,
                    # ... and by the evaluations the reference's loop needs (speculated Hessians are no credit)
                    "frac_needed": mk["fp64_gflop_needed_per_launch"] / mk["ms_isolated"] / 78.6,
                    "note": note + "; frac_step = the same flops / ms_per_step (what the pipelined step sustains); fp64 vector work (no "
                                   "MFMA instruction is issued: every pair term has its own 3x3 inverse), counted against the fp64 peak"}

    out = {
        "metric": "NDT scan-pair registrations/sec (100k pts, 0.5 m cells)",
        "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[2]: batch of %d independent 2D scan pairs per GPU, %d pts/scan, %.2f m cells, "
                               "map 100x100x1 m, range 30 m, n_neighbours 2, ITR_MAX 30, DELTA_SCORE 1e-6, 6-DoF, "
                               "grid build of both scans + D2D match per registration" % (B, NP, res),
                   "pairs_per_gpu": B, "points_per_scan": NP, "cell_m": res,
                   "entry": ("ndtgpu_mapset_build + ndtgpu_match_batch_device per step, streams and events driven by bench.py" if legacy
                             else "ONE ndtgpu_register_batch_device call per step (scans in HBM -> poses); reg.sync() ends the timed region"),
                   "pipeline": ("serial: one stream" if n_buf == 1 else
                                ("%d map sets / steps in flight, three streams, matcher launches on the CUs the builds leave" % n_buf if legacy else
                                 "%d internal map sets in flight: the grid builds of later steps run on two streams beside ONE running "
                                 "instance of the stream-fed matcher, which holds the share of the CUs the registrar measured on its "
                                 "first batch" % n_buf)),
                   "mean_cells_per_map": float((m_t.mean() + m_s.mean()) / 2),
                   "registrar": (reg.info() if reg is not None else None),
                   "batch_latency_ms": (batch_latency_ms if reg is not None else None)},
        "roofline": roofline, "kernels": kern,
        # SURVEY.md 8d (config 4): node maps and edges are separate units when node maps are reused across edges
        # north_star: scans/s and achieved HBM-bandwidth fraction (algorithmic bytes / time / 8 TB/s), per kernel
        "scans_per_s": 2 * value,
        "steady_state": steady,
        "hbm_fraction": {k: {"timed_region": kern[k]["GBps"] / HBM_PEAK_GBS,
                             "kernel_alone": kern[k]["algorithmic_bytes"] / kern[k]["ms_isolated"] / 1e6 / HBM_PEAK_GBS}
                         for k in kern},
        "collectives": {"process_group": "nccl (RCCL)" if use_dist else None, "world": world,
                        "forced_on_one_rank": bool(use_dist and world == 1),
                        "all_gather_into_tensor_calls": 2 * (args.steps + args.warmup + n_buf) if use_dist else 0,
                        # this rank's rows of the last step's gathered poses / results are the rows it computed
                        "gathered_rows_equal_local": (bool(torch.equal(last.gathered[0][rank * B:(rank + 1) * B], last.T16) and
                                                           torch.equal(last.gathered[1][rank * B:(rank + 1) * B], last.results))
                                                      if use_dist else None)},
        "nodes_per_s_build_only": world * 2 * B / (iso_build_ms * 1e-3),
        "edges_per_s_match_only_prebuilt_maps": world * B / (iso_match_ms * 1e-3),
    }

    # ---- CPU baseline + parity on a bounded sample (rank 0, N=1 only) ---------------------------
    # ---- single-pair latency of the other single-GPU configs, GPU side: the reference's call shape (build x2 + match, one
    #      host-synchronous call after the other).  Taken before the CPU legs (the oracle side of it follows them).
    def one_pair_gpu(fx, mv, T0, res_, size_, rng_, cap):
        ms = N.MapSet(res_, [0, 0, 0], size_, n_maps=2, max_cells=cap)
        scans = torch.stack([fx, mv]).contiguous()
        times = []
        for _ in range(24):
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            ms.build(scans, range_limit=rng_, stream=torch.cuda.current_stream())
            T, r = N.match_d2d(ms, 0, ms, 1, T0)
            times.append(time.perf_counter() - c0)
        times = sorted(times[3:])                      # the first calls load code objects and size the staging buffers
        best = times[len(times) // 2]                  # median of 21 (a minimum would hide a slow repeat call)
        return {"gpu_ms": 1e3 * best, "gpu_ms_min_max": [1e3 * times[0], 1e3 * times[-1]], "T": T, "r": r}

    lat_gpu = []
    if rank == 0 and world == 1 and not args.no_cpu:
        a2 = (fixed[0], moving[0], pr["T_init"][0].cpu().numpy(), res, size_m, rng_lim, 4096)
        lat_gpu.append(("configs[1] single 2D pair, %d pts, %.2f m cells (build x2 + match, host-synchronous call)" % (NP, res),
                        a2, one_pair_gpu(*a2)))
        p3 = synth.pair_3d(torch.tensor([1], device=dev), device=dev)
        a3 = (p3["fixed"][0], p3["moving"][0], p3["T_init"][0].cpu().numpy(), 0.25, [100.0, 100.0, 10.0], 70.0, 120000)
        lat_gpu.append(("configs[4] single 3D pair, 200000 pts, 0.25 m voxels, 100x100x10 m, 6-DoF", a3, one_pair_gpu(*a3)))
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_sample > 0:
        S = min(args.cpu_sample, B)
        f_h, m_h = fixed[:S].cpu().numpy(), moving[:S].cpu().numpy()
        Ti = pr["T_init"][:S].cpu().numpy()
        one, omp, To = cpu_baseline_c(f_h, m_h, Ti, res, size_m, rng_lim, 1e-6, 2, 30)
        max_dt = max_dr = 0.0
        for b in range(S):
            dt = float(np.linalg.norm(T_out[b][:3, 3] - To[b][:3, 3]))
            dr = float(2 * np.arcsin(min(1.0, np.linalg.norm(T_out[b][:3, :3] - To[b][:3, :3]) / (2 * np.sqrt(2)))))
            max_dt, max_dr = max(max_dt, dt), max(max_dr, dr)
        model, nproc = cpu_info()
        out["cpu_baseline"] = {"value": one["registrations_per_s"], "unit": "registrations/s", "cores": 1, "kind": "port",
                               "sample": "first %d of the %d pairs (same inputs, same parameters): oracle/ndt_oracle.c through the C "
                                         "driver oracle/cpu_baseline.c, gcc -O3 -march=native, taskset -c %d, 1 warm-up pass + median "
                                         "of %d passes (%.2f s per pass, min %.2f / max %.2f)" % (
                                             S, B, one["pinned_core"], one["reps"], one["median_pass_s"], one["min_pass_s"], one["max_pass_s"]),
                               "cpu_model": model, "nproc": nproc,
                               "all_cores": {"value": omp["registrations_per_s"], "unit": "registrations/s", "threads": omp["threads"],
                                             "note": "OpenMP over the pairs of the same sample (whole registrations in parallel), median of "
                                                     "%d passes; labelled figure, not the headline ratio" % omp["reps"]}}
        out["parity"] = {"pairs_checked": S, "max_dt_m": max_dt, "max_drot_rad": max_dr,
                         "tolerance": "1e-4 m / 1e-4 rad", "ok": bool(max_dt <= 1e-4 and max_dr <= 1e-4),
                         "note": "the timing driver is built -march=native (FMA contraction allowed): its poses agree with the GPU to the "
                                 "same bar as the strict oracle build the parity tests use"}
        out["speedup_vs_cpu_1thread"] = value / one["registrations_per_s"]
        out["speedup_vs_cpu_all_cores"] = value / omp["registrations_per_s"]
        # ---- the boundary hands over HOST clouds (what the reference call sites do): PCIe-inclusive rate, measured ------
        Bp = min(64, B)
        hs = N.MapSet(res, [0, 0, 0], size_m, n_maps=2 * Bp, max_cells=4096)
        both = np.concatenate([fixed[:Bp].cpu().numpy(), moving[:Bp].cpu().numpy()])
        Tp = pr["T_init"][:Bp].cpu().numpy()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            hs.build(both, range_limit=rng_lim)                                  # H2D of the raw scans + build (synchronous)
            N.match_batch(hs, np.arange(Bp), hs, np.arange(Bp) + Bp, Tp)         # H2D of poses, match, D2H of results
            best = min(best, time.perf_counter() - c0)
        # the asynchronous form: batch after batch of host clouds on one stream -- ndtgpu_mapset_build_host_async returns when
        # the host memory has been read, the registrations of batch k run on the device under the copies of batch k + 1;
        # poses and results come back with one copy per batch at the end
        hs2 = N.MapSet(res, [0, 0, 0], size_m, n_maps=2 * Bp, max_cells=4096)
        sets, n_batches = [hs, hs2], 6
        pst = torch.cuda.Stream(device=dev)
        idx_p = torch.arange(Bp, dtype=torch.int32, device=dev)
        idx_ps = idx_p + Bp
        Tp_cm = torch.as_tensor(np.ascontiguousarray(Tp.transpose(0, 2, 1)).reshape(Bp, 16), device=dev)
        T_b = [Tp_cm.clone() for _ in range(n_batches)]
        R_b = [torch.zeros((Bp, 64), dtype=torch.uint8, device=dev) for _ in range(n_batches)]
        asy = {}
        for mode in ("ring", "direct"):
            os.environ["NDTGPU_HOST_PIPE"] = "1" if mode == "ring" else "0"
            t_best = 1e9
            for _ in range(2):
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                with torch.cuda.stream(pst):
                    for k in range(n_batches):
                        ms_k = sets[k % 2]
                        ms_k.build(both, range_limit=rng_lim, stream=pst)        # host clouds, asynchronous form
                        T_b[k].copy_(Tp_cm)
                        binding.match_batch_device(ms_k, idx_p, ms_k, idx_ps, T_b[k], R_b[k], Bp, stream=pst)
                    T_host = [t.cpu() for t in T_b]
                pst.synchronize()
                t_best = min(t_best, time.perf_counter() - c0)
            asy[mode] = n_batches * Bp / t_best
        os.environ.pop("NDTGPU_HOST_PIPE", None)
        dT = float(np.abs(T_host[-1].numpy().reshape(Bp, 4, 4).transpose(0, 2, 1)[:, :3, 3] - T_out[:Bp, :3, 3]).max())
        out["pcie_inclusive"] = {"value": max(asy.values()), "unit": "registrations/s", "pairs": Bp,
                                 "synchronous_calls": Bp / best, "async_pinned_ring": asy["ring"], "async_one_copy": asy["direct"],
                                 "max_dt_vs_device_path_m": dT,
                                 "note": "host (pageable) clouds in, host poses out, %.1f MB H2D per registration; never the headline value. "
                                         "value = %d batches of %d pairs through ndtgpu_mapset_build_host_async + "
                                         "ndtgpu_match_batch_device on one stream (the copies of batch k + 1 run under the registrations "
                                         "of batch k; async_pinned_ring: chunks through the library's pinned ring, async_one_copy: one "
                                         "pageable copy per batch); synchronous_calls = one batch through the host-synchronous entries "
                                         "ndtgpu_mapset_build_host + ndtgpu_match_batch, the reference's call shape" % (
                                             2 * NP * 12 / 1e6, n_batches, Bp)}
        del hs2
        del hs
    # ---- single-pair latency of the other single-GPU configs (outside the timed region; rank 0, N=1): oracle side ---
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle as O
        lat = {}
        for name, (fx, mv, T0, res_, size_, rng_, cap), g in lat_gpu:
            if os.environ.get("BENCH_LAT_DEBUG"):
                again = one_pair_gpu(fx, mv, T0, res_, size_, rng_, cap)
                print("latency again, after the CPU legs: %.3f ms (before them %.3f)" % (again["gpu_ms"], g["gpu_ms"]), file=sys.stderr)
            T, r = g.pop("T"), g.pop("r")
            f_h, m_h = fx.cpu().numpy(), mv.cpu().numpy()
            c0 = time.perf_counter()
            ot = O.OracleMap(res_, [0, 0, 0], size_); ot.load_points(f_h, rng_); ot.compute_cells()
            os_ = O.OracleMap(res_, [0, 0, 0], size_); os_.load_points(m_h, rng_); os_.compute_cells()
            To, ro = O.match_d2d(ot, os_, T0)
            t_cpu = time.perf_counter() - c0
            dt = float(np.linalg.norm(T[:3, 3] - To[:3, 3]))
            dr = float(2 * np.arcsin(min(1.0, np.linalg.norm(T[:3, :3] - To[:3, :3]) / (2 * np.sqrt(2)))))
            g.update({"cpu_1thread_ms": 1e3 * t_cpu, "speedup": t_cpu / (1e-3 * g["gpu_ms"]),
                      "cells": [int(r["n_target"]), int(r["n_source"])], "iterations": int(r["iterations"]), "dt_m": dt, "drot_rad": dr})
            lat[name] = g
        out["single_pair_latency"] = lat
    # ---- the same configuration on a scene with ~2 k cells per map (SURVEY.md 8a/8d's size of a 2D map), labelled extra ----
    if rank == 0 and world == 1 and args.dense_pairs > 0 and B == 1024 and NP == 100000:
        for b in bufs:
            b.maps.close()
        if reg is not None:
            reg.close()
        del both, fixed, moving
        torch.cuda.empty_cache()
        out["dense_scene"] = dense_scene_leg(args, torch, N, binding, synth, dev, size_m, rng_lim, not args.no_cpu)
        torch.cuda.empty_cache()
        out["fuser_update"] = fuser_leg(torch, N, binding, synth, dev, not args.no_cpu)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
