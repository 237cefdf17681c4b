"""Sharding of independent scan-pair registrations over the GPUs of one node.

The path shards naturally: every NDTFeatureGraph::updateLinkUsingNDTRegistration call reads two
node maps and writes its own link (ndt_feature_graph.cpp:260-345; the loop :347-353 carries no
state), so there is NO data-path collective.  One process per GPU:

  * node maps are rebuilt on every rank (replicated build: 100 k points -> ~4 us of GPU time per
    scan, cheaper than shipping cell maps over xGMI and it removes the all-gather of phase B);
  * the edge list (identical on all ranks) is dealt block-cyclically, `chunk` edges at a time,
    because iteration counts vary per edge;
  * the only collective is the FINAL all-gather of the edge results {T 16 f64, result 64 B}
    (RCCL over xGMI on GPUs, gloo in the CPU tests): ~192 B/edge.
"""
import numpy as np


def shard_edges(n_edges, rank, world, chunk=256):
    """Indices (ascending) of the edges rank `rank` registers: chunks c with c % world == rank."""
    if world <= 1:
        return np.arange(n_edges, dtype=np.int64)
    ids = np.arange(n_edges, dtype=np.int64)
    return ids[((ids // chunk) % world) == rank]


def shard_sizes(n_edges, world, chunk=256):
    return [int(shard_edges(n_edges, r, world, chunk).shape[0]) for r in range(world)]


def gather_edge_results(T_local, res_local, n_edges, rank, world, chunk=256, group=None):
    """All-gathers per-edge results of the block-cyclic shards back into edge order.

    T_local: torch tensor [k,16] float64, res_local: torch tensor [k,64] uint8 (k = this rank's
    share, on the device the process group works on).  Returns (T [n_edges,16], res [n_edges,64])
    on every rank."""
    import torch
    import torch.distributed as dist
    if world <= 1:
        return T_local, res_local
    sizes = shard_sizes(n_edges, world, chunk)
    kmax = max(sizes)
    dev = T_local.device
    Tp = torch.zeros((kmax, 16), dtype=torch.float64, device=dev)
    Rp = torch.zeros((kmax, 64), dtype=torch.uint8, device=dev)
    k = T_local.shape[0]
    assert k == sizes[rank], "shard size mismatch: %d vs %d" % (k, sizes[rank])
    Tp[:k] = T_local
    Rp[:k] = res_local
    Tg = torch.empty((world * kmax, 16), dtype=torch.float64, device=dev)
    Rg = torch.empty((world * kmax, 64), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(Tg, Tp, group=group)
    dist.all_gather_into_tensor(Rg, Rp, group=group)
    T = torch.empty((n_edges, 16), dtype=torch.float64, device=dev)
    R = torch.empty((n_edges, 64), dtype=torch.uint8, device=dev)
    for r in range(world):
        ids = torch.as_tensor(shard_edges(n_edges, r, world, chunk), device=dev)
        T[ids] = Tg[r * kmax:r * kmax + sizes[r]]
        R[ids] = Rg[r * kmax:r * kmax + sizes[r]]
    return T, R


def register_sharded(n_edges, rank, world, register_fn, chunk=256, group=None):
    """One replay step of the multi-GPU path (bench.py --config 4 and tests/test_distributed.py run THIS function):
    this rank's block-cyclic share of the edge list goes to register_fn(mine) -> (T [k,16] float64, res [k,64] uint8)
    torch tensors on the process group's device -- the GPU matcher in bench.py, the CPU oracle under gloo in the tests --
    and the results of all ranks are all-gathered back into edge order (the only collective).
    Returns (mine, T [n_edges,16], res [n_edges,64])."""
    mine = shard_edges(n_edges, rank, world, chunk)
    T_local, res_local = register_fn(mine)
    T, R = gather_edge_results(T_local, res_local, n_edges, rank, world, chunk, group)
    return mine, T, R


def all_pairs(n_nodes):
    """NDTFeatureGraph::computeAllPossibleLinks enumeration order (ndt_feature_graph.cpp:395-405)."""
    i, j = np.triu_indices(n_nodes, k=1)
    return np.stack([i, j], axis=1).astype(np.int64)


def gate_links(edges, node_T, max_dist=1.0, max_angle=0.2, min_idx_dist=2):
    """The reference's candidate gates (NDTFeatureGraph::getValidLinks, ndt_feature_graph.cpp:527-556,
    defaults ndt_feature_graph_opt.cpp:49-52) applied to the odometry-predicted relative poses:
    keeps edges whose nodes are at least min_idx_dist apart in index and whose predicted relative
    translation / yaw are within the limits."""
    node_T = np.asarray(node_T, dtype=np.float64)
    keep = []
    for e, (i, j) in enumerate(edges):
        if abs(int(j) - int(i)) < min_idx_dist:
            continue
        rel = np.linalg.inv(node_T[i]) @ node_T[j]
        d = np.linalg.norm(rel[:3, 3])
        ang = abs(np.arctan2(rel[1, 0], rel[0, 0]))
        if d <= max_dist and ang <= max_angle:
            keep.append(e)
    return np.asarray(keep, dtype=np.int64)
