"""Seeded synthetic scan generators for the BASELINE.json configs.

The reference ships no data for this path (bags and .jff maps are missing,
ndt_feature/data/.MISSING_LARGE_BLOBS); its only synthetic-input recipe is the corridor of
ndt_feature/src/ndt_odom_debug.cpp:94-119 (range noise sigma 0.03 m, :49).  These generators
generalise that recipe: a 2D laser in a rectilinear room with box obstacles (configs 1-4)
and a Velodyne-style 64-ring sweep in a 3D hall (config 5).

Everything is a pure function of (seed, index) through a SplitMix64 counter hash written in
tensor ops, so the same code runs on the CPU (tests) and on the GPU (bench) and yields the
same scene; inputs are handed to both the HIP path and the CPU oracle as the same arrays.
"""
import math

import torch

_M64 = (1 << 64) - 1


def _i64(c):
    c &= _M64
    return c - (1 << 64) if c >= (1 << 63) else c


_GAMMA = _i64(0x9E3779B97F4A7C15)
_C1 = _i64(0xBF58476D1CE4E5B9)
_C2 = _i64(0x94D049BB133111EB)


def _lsr(x, k):
    return (x >> k) & ((1 << (64 - k)) - 1)


def _splitmix(x):
    z = x + _GAMMA
    z = (z ^ _lsr(z, 30)) * _C1
    z = (z ^ _lsr(z, 27)) * _C2
    return z ^ _lsr(z, 31)


def hash_uniform(seed, stream, idx):
    """U[0,1) doubles from (seed, stream, idx); idx is an int64 tensor, seed may be a tensor."""
    if not torch.is_tensor(seed):
        seed = torch.tensor(seed, dtype=torch.int64, device=idx.device)
    key = _splitmix(seed * 1000003 + stream)
    z = _splitmix(key ^ (idx * _GAMMA))
    return _lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def hash_normal(seed, stream, idx):
    u1 = hash_uniform(seed, stream, idx).clamp_min(1e-300)
    u2 = hash_uniform(seed, stream + 1, idx)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)


# ----------------------------------------------------------------------------------------
# 2D rooms (configs 1-4)
# ----------------------------------------------------------------------------------------

N_BOXES_2D = 6
# The "dense" scene (SURVEY.md 8a/8d sizes a 2D map at M ~ 1-3 k Gaussian cells; the plain room above yields ~370): the
# same hall with three thousand thin posts (rack legs, stanchions: 2-4 cm) -- thin enough that most of them stay visible from
# the sensor, each one a cell or two of its own.
N_BOXES_2D_DENSE = 3000


def room_2d(seeds, scene="room"):
    """Scene per seed: outer half extents [B,2] and boxes [B,K,4] = (cx, cy, hx, hy).  scene "room": 6 boxes of
    1-5 m; "dense": 3000 posts of 2-4 cm in the same room."""
    seeds = torch.as_tensor(seeds, dtype=torch.int64)
    dev = seeds.device
    B = seeds.shape[0]
    dense = scene == "dense"
    n_boxes = N_BOXES_2D_DENSE if dense else N_BOXES_2D
    k = torch.arange(n_boxes, dtype=torch.int64, device=dev)[None, :].expand(B, -1)
    s = seeds[:, None]
    two = torch.arange(2, dtype=torch.int64, device=dev)[None, :]
    half = 12.0 + 12.0 * hash_uniform(s, 11, two)                              # [B,2] in [12,24)
    cx = (2.0 * hash_uniform(s, 21, k) - 1.0) * (half[:, 0:1] - 3.0)
    cy = (2.0 * hash_uniform(s, 22, k) - 1.0) * (half[:, 1:2] - 3.0)
    if dense:
        hx = 0.01 + 0.01 * hash_uniform(s, 23, k)
        hy = 0.01 + 0.01 * hash_uniform(s, 24, k)
    else:
        hx = 0.5 + 2.0 * hash_uniform(s, 23, k)
        hy = 0.5 + 2.0 * hash_uniform(s, 24, k)
    # keep a free area around the sensor poses (|x|,|y| < ~2): push close boxes outwards
    close = (cx.abs() < 4.5 + hx) & (cy.abs() < 4.5 + hy)
    cx = torch.where(close, torch.where(cx >= 0, cx + 7.0, cx - 7.0), cx)
    return half, torch.stack([cx, cy, hx, hy], dim=-1)


def scan_2d(seeds, poses, n_points, noise_sigma=0.03, z_jitter=0.02, r_max=30.0, r_min=0.5,
            noise_stream=0, chunk_bytes=256 << 20, scene="room"):
    """Laser scans of room_2d(seed) from sensor poses (x, y, yaw) -> float32 [B, n_points, 3].

    Beams uniform over 2*pi in the sensor frame; range noise N(0, sigma^2)
    (ndt_odom_debug.cpp:49); r > r_max or r < r_min dropped as NaN (sensor_range / min range,
    gustav_laser_tf.launch:22-23); z = z_jitter * U[0,1) (laser_variance_z,
    publish_graph_message.cpp:1375-1381).
    """
    seeds = torch.as_tensor(seeds, dtype=torch.int64)
    dev = seeds.device
    poses = torch.as_tensor(poses, dtype=torch.float64, device=dev)
    B = seeds.shape[0]
    half, boxes = room_2d(seeds, scene)
    out = torch.empty((B, n_points, 3), dtype=torch.float32, device=dev)
    nseg = 4 * (1 + boxes.shape[1])
    per_scan = n_points * nseg * 8 * 6
    bchunk = max(1, int(chunk_bytes // max(per_scan, 1)))
    idx = torch.arange(n_points, dtype=torch.int64, device=dev)
    for b0 in range(0, B, bchunk):
        b1 = min(B, b0 + bchunk)
        sd = seeds[b0:b1, None]
        ox = poses[b0:b1, 0:1]
        oy = poses[b0:b1, 1:2]
        yaw = poses[b0:b1, 2:3]
        phi = (idx[None, :].to(torch.float64) + 0.5) * (2.0 * math.pi / n_points) - math.pi   # sensor frame
        ang = phi + yaw
        dx = torch.cos(ang)
        dy = torch.sin(ang)
        # rectangles: outer room + boxes -> (cx, cy, hx, hy) [b, R]
        zero = torch.zeros_like(half[b0:b1, 0:1])
        rect = torch.cat([torch.stack([zero, zero, half[b0:b1, 0:1], half[b0:b1, 1:2]], dim=-1),
                          boxes[b0:b1]], dim=1)                                   # [b, R, 4]
        xs = torch.stack([rect[..., 0] - rect[..., 2], rect[..., 0] + rect[..., 2]], dim=-1)  # [b,R,2]
        ys = torch.stack([rect[..., 1] - rect[..., 3], rect[..., 1] + rect[..., 3]], dim=-1)
        big = 1e9
        # vertical walls x = xs, y in [ylo, yhi]
        tx = (xs.reshape(b1 - b0, 1, -1) - ox[:, :, None]) / dx[:, :, None]       # [b,N,2R]
        yhit = oy[:, :, None] + tx * dy[:, :, None]
        ylo = ys[..., 0].repeat_interleave(2, dim=1)[:, None, :]
        yhi = ys[..., 1].repeat_interleave(2, dim=1)[:, None, :]
        okx = (tx > 1e-9) & (yhit >= ylo) & (yhit <= yhi)
        tx = torch.where(okx, tx, torch.full_like(tx, big))
        r = tx.min(dim=-1).values
        del tx, yhit, okx
        ty = (ys.reshape(b1 - b0, 1, -1) - oy[:, :, None]) / dy[:, :, None]
        xhit = ox[:, :, None] + ty * dx[:, :, None]
        xlo = xs[..., 0].repeat_interleave(2, dim=1)[:, None, :]
        xhi = xs[..., 1].repeat_interleave(2, dim=1)[:, None, :]
        oky = (ty > 1e-9) & (xhit >= xlo) & (xhit <= xhi)
        ty = torch.where(oky, ty, torch.full_like(ty, big))
        r = torch.minimum(r, ty.min(dim=-1).values)
        del ty, xhit, oky
        r = r + noise_sigma * hash_normal(sd, 31 + 10 * noise_stream, idx[None, :])
        bad = (r > r_max) | (r < r_min)
        z = z_jitter * hash_uniform(sd, 41 + 10 * noise_stream, idx[None, :])
        px = torch.where(bad, torch.full_like(r, float("nan")), r * torch.cos(phi))
        py = torch.where(bad, torch.full_like(r, float("nan")), r * torch.sin(phi))
        pz = torch.where(bad, torch.full_like(r, float("nan")), z)
        out[b0:b1, :, 0] = px.to(torch.float32)
        out[b0:b1, :, 1] = py.to(torch.float32)
        out[b0:b1, :, 2] = pz.to(torch.float32)
    return out


def pose2d_to_T(pose):
    """(x, y, yaw) [B,3] -> 4x4 [B,4,4] float64 (Translation * Rz)."""
    pose = torch.as_tensor(pose, dtype=torch.float64)
    B = pose.shape[0]
    T = torch.zeros((B, 4, 4), dtype=torch.float64, device=pose.device)
    c, s = torch.cos(pose[:, 2]), torch.sin(pose[:, 2])
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = c, -s, s, c
    T[:, 2, 2] = 1.0
    T[:, 3, 3] = 1.0
    T[:, 0, 3], T[:, 1, 3] = pose[:, 0], pose[:, 1]
    return T


# offset between the two sensor poses and the odometry-like perturbation of the initial guess
# 
PAIR_OFFSET_2D = (0.30, 0.10, math.radians(3.0))
GUESS_PERTURB_2D = (0.10, -0.05, math.radians(1.0))


def pair_2d(seeds, n_points, device="cpu", **kw):
    """Scan pairs for configs 1-3: returns dict(fixed [B,N,3] f32, moving [B,N,3] f32,
    T_gt [B,4,4], T_init [B,4,4]).  The fixed scan is taken at a seed-dependent pose near the
    origin, the moving scan at fixed * PAIR_OFFSET_2D; match(target=fixed, source=moving, T)
    must recover T_gt = offset."""
    seeds = torch.as_tensor(seeds, dtype=torch.int64, device=device)
    B = seeds.shape[0]
    three = torch.arange(3, dtype=torch.int64, device=device)[None, :]
    u = hash_uniform(seeds[:, None], 51, three)
    pose_a = torch.stack([(u[:, 0] - 0.5) * 2.0, (u[:, 1] - 0.5) * 2.0, (u[:, 2] - 0.5) * 2.0 * math.pi], dim=-1)
    off = torch.tensor(PAIR_OFFSET_2D, dtype=torch.float64, device=device)[None, :].expand(B, -1)
    Ta = pose2d_to_T(pose_a)
    Toff = pose2d_to_T(off)
    Tb = Ta @ Toff
    pose_b = torch.stack([Tb[:, 0, 3], Tb[:, 1, 3], pose_a[:, 2] + off[:, 2]], dim=-1)
    fixed = scan_2d(seeds, pose_a, n_points, noise_stream=0, **kw)
    moving = scan_2d(seeds, pose_b, n_points, noise_stream=1, **kw)
    g = torch.tensor([PAIR_OFFSET_2D[i] + GUESS_PERTURB_2D[i] for i in range(3)], dtype=torch.float64,
                     device=device)[None, :].expand(B, -1)
    return dict(fixed=fixed, moving=moving, T_gt=Toff, T_init=pose2d_to_T(g))


# ----------------------------------------------------------------------------------------
# 3D hall (config 5)
# ----------------------------------------------------------------------------------------

N_BOXES_3D = 10
PAIR_OFFSET_3D = (0.5, 0.2, 0.05, math.radians(1.0), math.radians(-1.0), math.radians(4.0))
GUESS_PERTURB_3D = (0.10, -0.05, 0.02, math.radians(0.3), math.radians(-0.2), math.radians(1.0))


def pose6_to_T(p):
    """(x,y,z,rx,ry,rz) [B,6] -> [B,4,4]: Translation*Rx*Ry*Rz
    (ndt_matcher_d2d_fusion.h:1036-1039)."""
    p = torch.as_tensor(p, dtype=torch.float64)
    B = p.shape[0]
    cx, sx = torch.cos(p[:, 3]), torch.sin(p[:, 3])
    cy, sy = torch.cos(p[:, 4]), torch.sin(p[:, 4])
    cz, sz = torch.cos(p[:, 5]), torch.sin(p[:, 5])
    one, zero = torch.ones_like(cx), torch.zeros_like(cx)
    Rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=-1).reshape(B, 3, 3)
    Ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=-1).reshape(B, 3, 3)
    Rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=-1).reshape(B, 3, 3)
    T = torch.zeros((B, 4, 4), dtype=torch.float64, device=p.device)
    T[:, :3, :3] = Rx @ Ry @ Rz
    T[:, :3, 3] = p[:, :3]
    T[:, 3, 3] = 1.0
    return T


def scan_3d(seeds, T_sensor, rings=64, azimuths=3125, noise_sigma=0.02, r_max=70.0, r_min=1.5,
            noise_stream=0):
    """Velodyne-style sweep inside a hall (floor z=-1.8, ceiling, 4 walls, axis-aligned boxes).
    T_sensor [B,4,4] world<-sensor.  Returns float32 [B, rings*azimuths, 3] in the sensor frame
    (NaN where out of range; max/min range as ndt_graph_offline.cpp:185-186)."""
    seeds = torch.as_tensor(seeds, dtype=torch.int64)
    dev = seeds.device
    T_sensor = torch.as_tensor(T_sensor, dtype=torch.float64, device=dev)
    B = seeds.shape[0]
    N = rings * azimuths
    idx = torch.arange(N, dtype=torch.int64, device=dev)
    ring = (idx // azimuths).to(torch.float64)
    az = (idx % azimuths).to(torch.float64)
    elev = math.radians(-24.8) + ring * (math.radians(26.8) / (rings - 1))
    theta = (az + 0.5) * (2.0 * math.pi / azimuths) - math.pi
    d_s = torch.stack([torch.cos(elev) * torch.cos(theta), torch.cos(elev) * torch.sin(theta), torch.sin(elev)],
                      dim=-1)                                                      # [N,3] sensor frame
    out = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    kk = torch.arange(N_BOXES_3D, dtype=torch.int64, device=dev)
    for b in range(B):
        sd = seeds[b]
        R = T_sensor[b, :3, :3]
        o = T_sensor[b, :3, 3]
        d = d_s @ R.T                                                               # [N,3] world
        hw = 18.0 + 14.0 * hash_uniform(sd, 61, torch.arange(2, dtype=torch.int64, device=dev))
        lo = torch.stack([-hw[0], -hw[1], torch.tensor(-1.8, dtype=torch.float64, device=dev)])
        hi = torch.stack([hw[0], hw[1], torch.tensor(6.0, dtype=torch.float64, device=dev)])
        inv = 1.0 / d
        # inside the hall: exit distance = min over axes of the far slab
        t_far = torch.maximum((lo[None, :] - o[None, :]) * inv, (hi[None, :] - o[None, :]) * inv)
        r = t_far.min(dim=-1).values
        # boxes: entry distance (slab method)
        bc = torch.stack([(2.0 * hash_uniform(sd, 71, kk) - 1.0) * (hw[0] - 3.0),
                          (2.0 * hash_uniform(sd, 72, kk) - 1.0) * (hw[1] - 3.0)], dim=-1)
        close = (bc[:, 0].abs() < 6.0) & (bc[:, 1].abs() < 6.0)
        bc[:, 0] = torch.where(close, torch.where(bc[:, 0] >= 0, bc[:, 0] + 9.0, bc[:, 0] - 9.0), bc[:, 0])
        bh = torch.stack([0.5 + 2.0 * hash_uniform(sd, 73, kk), 0.5 + 2.0 * hash_uniform(sd, 74, kk),
                          0.5 + 2.5 * hash_uniform(sd, 75, kk)], dim=-1)            # half x, half y, height
        blo = torch.stack([bc[:, 0] - bh[:, 0], bc[:, 1] - bh[:, 1], torch.full_like(bh[:, 2], -1.8)], dim=-1)
        bhi = torch.stack([bc[:, 0] + bh[:, 0], bc[:, 1] + bh[:, 1], -1.8 + bh[:, 2]], dim=-1)
        t1 = (blo[None, :, :] - o[None, None, :]) * inv[:, None, :]
        t2 = (bhi[None, :, :] - o[None, None, :]) * inv[:, None, :]
        tn = torch.minimum(t1, t2).max(dim=-1).values                              # [N,K]
        tf = torch.maximum(t1, t2).min(dim=-1).values
        hit = (tn <= tf) & (tn > 1e-9)
        tn = torch.where(hit, tn, torch.full_like(tn, 1e9))
        r = torch.minimum(r, tn.min(dim=-1).values)
        r = r + noise_sigma * hash_normal(sd, 81 + 10 * noise_stream, idx)
        bad = (r > r_max) | (r < r_min)
        p = d_s * r[:, None]
        p = torch.where(bad[:, None], torch.full_like(p, float("nan")), p)
        out[b] = p.to(torch.float32)
    return out


def pair_3d(seeds, rings=64, azimuths=3125, device="cpu"):
    seeds = torch.as_tensor(seeds, dtype=torch.int64, device=device)
    B = seeds.shape[0]
    Ta = torch.eye(4, dtype=torch.float64, device=device)[None].repeat(B, 1, 1)
    off = torch.tensor(PAIR_OFFSET_3D, dtype=torch.float64, device=device)[None, :].expand(B, -1)
    Toff = pose6_to_T(off)
    Tb = Ta @ Toff
    fixed = scan_3d(seeds, Ta, rings, azimuths, noise_stream=0)
    moving = scan_3d(seeds, Tb, rings, azimuths, noise_stream=1)
    g = torch.tensor([PAIR_OFFSET_3D[i] + GUESS_PERTURB_3D[i] for i in range(6)], dtype=torch.float64,
                     device=device)[None, :].expand(B, -1)
    return dict(fixed=fixed, moving=moving, T_gt=Toff, T_init=pose6_to_T(g))
