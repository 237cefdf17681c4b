"""Sharding of independent scan-pair registrations over the GPUs of one node.

The path shards naturally: every NDTFeatureGraph::updateLinkUsingNDTRegistration call reads two
node maps and writes its own link (ndt_feature_graph.cpp:260-345; the loop :347-353 carries no
state), so there is NO data-path collective.  One process per GPU:

  * node maps are built data-parallel, node k on rank k mod world (SURVEY.md 8e phase A), packed into fixed-stride
    exchange records on the device (ndtgpu_mapset_pack_cells_device: the Gaussian cells in slot order + the occupancy of
    every cell) and all-gathered ONCE (phase B); every rank unpacks all of them (cells, rank map, counters).  A fused node
    map costs ~28 us of GPU time (ten ray-traced scans): replicating that on 8 ranks capped the replay at 3.5x;
    plain one-scan maps (configs 1-3, 5) need no exchange at all -- each rank builds the scans of its own pairs;
  * the edge list (identical on all ranks) is dealt block-cyclically, `chunk` edges at a time,
    because iteration counts vary per edge;
  * the only collective is the FINAL all-gather of the edge results {T 16 f64, result 64 B}
    (RCCL over xGMI on GPUs, gloo in the CPU tests): ~192 B/edge.
"""
import os

import numpy as np


def collectives_forced():
    """NDTGPU_FORCE_COLLECTIVES=1: a ONE-rank run still creates the process group and goes through every collective of the
    path (both all-gathers of the edge results, the exchange of the packed node maps) instead of taking the world == 1
    short cuts -- the RCCL calls, their streams and their buffer layouts are executed on whatever single GPU is there
    (tests/test_gpu_rccl.py), before an 8-GPU node ever runs them."""
    return os.environ.get("NDTGPU_FORCE_COLLECTIVES", "0") not in ("", "0")


def _collective(world):
    if world > 1:
        return True
    if not collectives_forced():
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


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
    if not _collective(world):
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


def shard_nodes(n_nodes, rank, world):
    """Nodes whose maps rank `rank` builds: k with k % world == rank (SURVEY.md 8e phase A)."""
    return np.arange(rank, n_nodes, max(1, world), dtype=np.int64)


def records_to_node_order(gathered, n_nodes, world):
    """The all-gathered records [world * n_max, stride] (rank-major, every rank's share padded to n_max = ceil(n / world))
    in NODE order [n_nodes, stride]: record i of rank r is node i * world + r."""
    n_max = (n_nodes + world - 1) // world
    stride = gathered.shape[1]
    return gathered.view(world, n_max, stride).transpose(0, 1).reshape(world * n_max, stride)[:n_nodes].contiguous()


def exchange_node_maps(packed_local, n_nodes, rank, world, group=None):
    """Phase B: ONE all-gather of the exchange records.  packed_local: torch uint8 [n_local, stride], record i = node
    rank + i * world (ndtgpu_mapset_pack_cells_device on GPUs; any producer of the same layout in the CPU tests).
    Returns uint8 [n_nodes, stride] in NODE order on every rank -- what ndtgpu_mapset_unpack_cells_device installs with
    one call."""
    import torch
    import torch.distributed as dist
    if not _collective(world):
        return packed_local[:n_nodes]
    n_max = (n_nodes + world - 1) // world
    stride = packed_local.shape[1]
    mine = packed_local
    if mine.shape[0] != n_max:                              # the last ranks hold one node less: pad to equal shares
        pad = torch.zeros((n_max, stride), dtype=torch.uint8, device=packed_local.device)
        pad[:mine.shape[0]] = mine
        mine = pad
    gathered = torch.empty((world * n_max, stride), dtype=torch.uint8, device=packed_local.device)
    dist.all_gather_into_tensor(gathered, mine.contiguous(), group=group)
    return records_to_node_order(gathered, n_nodes, world)


# the exchange record of include/ndtgpu.h (ndtgpu_packed_header + ndtgpu_cell_record) as NumPy dtypes: host-side views of
# packed buffers (tests, tools); the device side is csrc/ndt_pack.hip
PACKED_HEADER_DTYPE = np.dtype([("n_cells", "<u4"), ("flags", "<u4"), ("n_dropped", "<u4"), ("cells_cap", "<u4")])
PACKED_CELL_DTYPE = np.dtype([("mean", "<f8", (3,)), ("cov", "<f8", (6,)), ("n", "<u4"), ("slot", "<u4")])
assert PACKED_HEADER_DTYPE.itemsize == 16 and PACKED_CELL_DTYPE.itemsize == 80


PACKED_OCC_HEAD_DTYPE = np.dtype([("n_occ", "<u4"), ("occ_cap", "<u4")])      # sparse occupancy block: head, then pairs
PACKED_OCC_DTYPE = np.dtype([("slot", "<u4"), ("occ", "<f4")])


def record_bytes(cells_cap, slots=0, occ_cap=None):
    """ndtgpu_mapset_pack_bytes: bytes of one record (slots > 0: with the occupancy of every cell); occ_cap:
    ndtgpu_mapset_pack_bytes_sparse (the (slot, occupancy) pairs of up to occ_cap cells with a reading instead)."""
    if occ_cap is not None:
        return (16 + 80 * int(cells_cap) + 8 + 8 * int(occ_cap) + 15) // 16 * 16
    return (16 + 80 * int(cells_cap) + 4 * int(slots) + 15) // 16 * 16


def sparse_occupancy_of_record(rec):
    """(slots, values) of the sparse occupancy block of ONE exchange record (uint8 array), or None when the record carries
    its occupancies densely or not at all (flags bit 2 clear)."""
    rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1)
    h = rec[:16].view(PACKED_HEADER_DTYPE)[0]
    if not (int(h["flags"]) & 4):
        return None
    at = 16 + 80 * int(h["cells_cap"])
    oh = rec[at:at + 8].view(PACKED_OCC_HEAD_DTYPE)[0]
    pairs = rec[at + 8:at + 8 + 8 * int(oh["n_occ"])].view(PACKED_OCC_DTYPE)
    return pairs["slot"].copy(), pairs["occ"].copy()


def record_from_cells(mean, cov, idx, npts, cells_per_axis, cells_cap, n_dropped=0):
    """One exchange record (uint8 [record_bytes]) from exported cells: mean [n,3], cov [n,3,3], idx [n,3] LazyGrid indices,
    npts [n]; the cells are put in slot order like the device's."""
    sx, sy, sz = (int(v) for v in cells_per_axis)
    idx = np.asarray(idx, dtype=np.int64).reshape(-1, 3)
    slot = (idx[:, 0] * sy + idx[:, 1]) * sz + idx[:, 2]
    order = np.argsort(slot, kind="stable")
    n = min(len(order), int(cells_cap))
    rec = np.zeros(record_bytes(cells_cap), dtype=np.uint8)
    h = rec[:16].view(PACKED_HEADER_DTYPE)
    h["n_cells"], h["flags"], h["n_dropped"], h["cells_cap"] = n, (1 if len(order) > cells_cap else 0), n_dropped, cells_cap
    c = rec[16:16 + 80 * int(cells_cap)].view(PACKED_CELL_DTYPE)
    cov = np.asarray(cov, dtype=np.float64).reshape(-1, 3, 3)
    o = order[:n]
    c["mean"][:n] = np.asarray(mean, dtype=np.float64).reshape(-1, 3)[o]
    c["cov"][:n] = np.stack([cov[o, 0, 0], cov[o, 0, 1], cov[o, 0, 2], cov[o, 1, 1], cov[o, 1, 2], cov[o, 2, 2]], axis=1)
    c["n"][:n] = np.asarray(npts).reshape(-1)[o]
    c["slot"][:n] = slot[o]
    return rec


def cells_from_record(rec):
    """(mean [n,3], cov [n,3,3], slot [n], npts [n], flags) of one exchange record."""
    rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1)
    h = rec[:16].view(PACKED_HEADER_DTYPE)[0]
    n, cap = int(h["n_cells"]), int(h["cells_cap"])
    c = rec[16:16 + 80 * cap].view(PACKED_CELL_DTYPE)[:n]
    v = c["cov"]
    cov = np.stack([v[:, 0], v[:, 1], v[:, 2], v[:, 1], v[:, 3], v[:, 4], v[:, 2], v[:, 4], v[:, 5]], axis=1).reshape(-1, 3, 3)
    return c["mean"].copy(), cov, c["slot"].copy(), c["n"].copy(), int(h["flags"])


def phase_model(build_all_ms, match_all_ms, n_nodes, record_bytes, n_edges, worlds=(1, 2, 4, 8), link_GBps=153.0, links=7):
    """Per-phase time model of one replay step on `world` GPUs of one node, from the one-rank measurements: node builds
    and edge registrations divide by world; the exchange is an all-gather of n_nodes records over point-to-point xGMI
    (each rank sends its share to the other world - 1 ranks over its own links, ~153 GB/s per link), the final gather one of
    192 B per edge.  A model, not a measurement: no multi-GPU node was available to any round."""
    out = {}
    for w in worlds:
        per_rank_out = record_bytes * n_nodes / w * (w - 1)          # bytes a rank puts on its links
        bw = link_GBps * 1e9 * min(links, max(1, w - 1))
        gather_ms = 0.0 if w == 1 else 1e3 * per_rank_out / bw
        edge_ms = 0.0 if w == 1 else 1e3 * (192.0 * n_edges / w * (w - 1)) / bw
        total = build_all_ms / w + gather_ms + match_all_ms / w + edge_ms
        out[str(w)] = {"build_ms": build_all_ms / w, "exchange_ms": gather_ms, "match_ms": match_all_ms / w,
                       "edge_gather_ms": edge_ms, "step_ms": total,
                       "speedup": (build_all_ms + match_all_ms) / total}
    return out


def all_pairs(n_nodes):
    """NDTFeatureGraph::computeAllPossibleLinks enumeration order (ndt_feature_graph.cpp:395-405)."""
    i, j = np.triu_indices(n_nodes, k=1)
    return np.stack([i, j], axis=1).astype(np.int64)


def valid_links(edges, T_links, node_T, scores=None, max_score=0.1, max_dist=1.0, max_angle=0.2, min_idx_dist=2):
    """NDTFeatureGraph::getValidLinks (ndt_feature_graph.cpp:527-556) with the defaults of ndt_feature_graph_opt.cpp:49-52,
    the filter the reference applies to the REGISTERED links before it optimises: a link (ref, mov, T) stays when its
    score <= max_score (skipped when `scores` is None), its nodes are at least min_idx_dist indices apart and the pose it
    predicts for the moving node, node_T[ref] * T, lies within max_dist / max_angle of the node's own pose
    (distanceBetweenAffine3d, utils.h:43-48: norm of the translation of own^-1 * pred, and |getRobustYawFromAffine3d| =
    acos of the (0, 0) entry of its rotation -- the angle of the rotated x axis IN THE XY PLANE, not the 3D rotation angle, and
    unclipped as upstream: an entry a rounding above 1 gives NaN, NaN < max_angle is false, the link is rejected).
    Returns the indices of the links that stay."""
    edges = np.asarray(edges)
    T_links = np.asarray(T_links, dtype=np.float64).reshape(-1, 4, 4)
    node_T = np.asarray(node_T, dtype=np.float64)
    pred = np.einsum("eij,ejk->eik", node_T[edges[:, 0]], T_links)
    own = node_T[edges[:, 1]]
    dist = np.linalg.norm(pred[:, :3, 3] - own[:, :3, 3], axis=1)
    r00 = np.einsum("ej,ej->e", own[:, :3, 0], pred[:, :3, 0])          # (own_R^T pred_R)(0, 0)
    with np.errstate(invalid="ignore"):
        ang = np.abs(np.arccos(r00))
    keep = (np.abs(edges[:, 1].astype(np.int64) - edges[:, 0].astype(np.int64)) >= min_idx_dist) & (dist < max_dist) & (ang < max_angle)
    if scores is not None:
        keep &= np.asarray(scores) <= max_score
    return np.nonzero(keep)[0].astype(np.int64)


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
