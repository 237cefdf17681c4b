"""N>1 path on CPU: world_size-2 gloo processes build their share of the node maps (node k on rank k mod world), exchange
them as the device's packed records with ONE all-gather, shard the edge list block-cyclically, register their share (the
CPU oracle stands in for the GPU builder and matcher -- tests may use it as the checker), all-gather the edge results and
must reproduce the single-process answer exactly."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_edges_partition():
    from ndt_feature_graph_amd import distributed as D
    for n, w, c in [(1000, 2, 256), (10, 4, 3), (0, 2, 8), (777, 8, 16), (5, 2, 256)]:
        parts = [D.shard_edges(n, r, w, c) for r in range(w)]
        allids = np.sort(np.concatenate(parts)) if n else np.zeros(0, np.int64)
        assert np.array_equal(allids, np.arange(n))
        assert D.shard_sizes(n, w, c) == [len(p) for p in parts]
    assert np.array_equal(D.shard_edges(10, 0, 1), np.arange(10))


def test_all_pairs_and_gates():
    from ndt_feature_graph_amd import distributed as D
    e = D.all_pairs(5)
    assert e.shape == (10, 2) and tuple(e[0]) == (0, 1) and tuple(e[-1]) == (3, 4)     # graph.cpp:395-405 order
    T = np.stack([np.eye(4) for _ in range(5)])
    for k in range(5):
        T[k, 0, 3] = 0.4 * k
    keep = D.gate_links(e, T, max_dist=1.0, max_angle=0.2, min_idx_dist=2)
    assert [tuple(e[k]) for k in keep] == [(0, 2), (1, 3), (2, 4)]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle as O
    from ndt_feature_graph_amd import distributed as D, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_nodes, chunk = 5, 2
    # node k = scan of room seed 77 from a pose 0.25 m further along x.  Phase A: this rank builds nodes k % world == rank;
    # phase B: their cells travel as the exchange records of include/ndtgpu.h through ONE all-gather; every rank installs
    # all of them (bench.py --config 4 does the same with ndtgpu_mapset_pack_cells_device / _unpack_cells_device)
    poses = np.array([[0.25 * k, 0.05 * k, 0.01 * k] for k in range(n_nodes)])
    scans = synth.scan_2d([77] * n_nodes, poses, 3000).numpy()
    cap, grid = 512, (100, 100, 1)
    recs = []
    for k in D.shard_nodes(n_nodes, rank, world):
        m = O.OracleMap(1.0, [0, 0, 0], [100, 100, 1]); m.load_points(scans[k], 30.0); m.compute_cells()
        recs.append(D.record_from_cells(*m.export_cells(), grid, cap))
    packed = torch.from_numpy(np.stack(recs)) if recs else torch.zeros((0, D.record_bytes(cap)), dtype=torch.uint8)
    allrec = D.exchange_node_maps(packed, n_nodes, rank, world).numpy()
    assert allrec.shape == (n_nodes, D.record_bytes(cap))
    maps = []
    for k in range(n_nodes):
        mean, cov, slot, npts, flags = D.cells_from_record(allrec[k])
        assert flags == 0 and np.all(np.diff(slot.astype(np.int64)) > 0)            # slot order, nothing cut
        m = O.OracleMap(1.0, [0, 0, 0], [100, 100, 1]); m.set_cells(mean, cov)
        assert m.num_cells() == len(slot)
        maps.append(m)
    edges = D.all_pairs(n_nodes)
    node_T = synth.pose2d_to_T(poses).numpy()

    def register(mine):                       # the CPU oracle stands in for the GPU matcher of bench.py --config 4
        T_loc = np.zeros((len(mine), 16))
        R_loc = np.zeros((len(mine), 64), np.uint8)
        for q, e in enumerate(mine):
            i, j = edges[e]
            T0 = np.linalg.inv(node_T[i]) @ node_T[j]
            T0[0, 3] += 0.03
            T, r = O.match_d2d(maps[i], maps[j], T0)
            T_loc[q] = T.T.reshape(-1)
            R_loc[q, :4] = np.frombuffer(np.int32(r["iterations"]).tobytes(), np.uint8)
        return torch.from_numpy(T_loc), torch.from_numpy(R_loc)
    # the replay step bench.py --config 4 runs on every rank: shard, register, all-gather
    mine, Tg, Rg = D.register_sharded(len(edges), rank, world, register, chunk)
    assert np.array_equal(mine, D.shard_edges(len(edges), rank, world, chunk))
    np.save(os.path.join(out_dir, "T_w%d_r%d.npy" % (world, rank)), Tg.numpy())
    np.save(os.path.join(out_dir, "R_w%d_r%d.npy" % (world, rank)), Rg.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_valid_links_restates_getValidLinks():
    """NDTFeatureGraph::getValidLinks (ndt_feature_graph.cpp:527-556; defaults ndt_feature_graph_opt.cpp:49-52): score <= 0.1,
    nodes >= 2 indices apart, the moving node's pose predicted through the link within 1.0 m / 0.2 rad of its own pose --
    strict inequalities on distance and angle, like upstream."""
    from ndt_feature_graph_amd import distributed as D

    def pose(x, y, yaw):
        T = np.eye(4)
        T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
        T[0, 3], T[1, 3] = x, y
        return T
    nodes = np.stack([pose(0, 0, 0), pose(2, 0, 0.1), pose(4, 0, 0.2), pose(6, 1, 0.3)])
    exact = lambda i, j: np.linalg.inv(nodes[i]) @ nodes[j]
    edges = np.array([[0, 1], [0, 2], [1, 3], [0, 3], [0, 2], [1, 3]])
    T = np.stack([exact(0, 1),                                  # consecutive nodes: dropped by min_idx_dist
                  exact(0, 2),                                  # the odometry itself: kept
                  exact(1, 3) @ pose(0.5, 0.0, 0.0),            # 0.5 m off: kept
                  exact(0, 3) @ pose(1.001, 0.0, 0.0),          # 1.001 m off: dropped
                  exact(0, 2) @ pose(0.0, 0.0, 0.25),           # 0.25 rad off: dropped
                  exact(1, 3) @ pose(0.0, 0.3, 0.19)])          # 0.3 m and 0.19 rad off: kept
    assert D.valid_links(edges, T, nodes).tolist() == [1, 2, 5]
    assert D.valid_links(edges, T, nodes, scores=[0, 0.05, 0.2, 0, 0, 0.1]).tolist() == [1, 5]      # score > 0.1 dropped
    assert D.valid_links(edges, T, nodes, min_idx_dist=1).tolist() == [0, 1, 2, 5]
    assert D.valid_links(edges[:0], T[:0], nodes).tolist() == []


def test_two_rank_gloo_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for world in (1, 2):
        for attempt in range(3):          # the probed port can be taken again before the ranks bind it: retry
            port = _free_port()
            procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
            for p in procs:
                p.start()
            for p in procs:
                p.join(240)
            if all(p.exitcode == 0 for p in procs):
                break
            for p in procs:
                if p.is_alive():
                    p.kill()
        assert all(p.exitcode == 0 for p in procs)
    T1 = np.load(tmp_path / "T_w1_r0.npy")
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("T_w2_r%d.npy" % r)), T1)         # every rank holds all edges
        assert np.array_equal(np.load(tmp_path / ("R_w2_r%d.npy" % r)), np.load(tmp_path / "R_w1_r0.npy"))
    assert T1.shape == (10, 16) and np.all(np.abs(T1[:, 15] - 1.0) < 1e-15)
