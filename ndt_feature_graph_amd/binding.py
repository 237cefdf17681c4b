"""ctypes host layer over libndtgpu.so (include/ndtgpu.h)."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_HERE, "libndtgpu.so")
_SOURCES = ["ndt_build.hip", "ndt_build_flat.hip", "ndt_match.hip", "ndt_fuse.hip", "ndt_pack.hip", "ndt_fuser.hip", "ndtgpu_api.hip"]

STATUS = {0: "OK", -1: "ERR_INVALID", -2: "ERR_HIP", -3: "ERR_NO_DEVICE", -4: "ERR_CAPACITY", -5: "ERR_ALLOC"}


class NdtGpuError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("ndtgpu %s (%d): %s" % (STATUS.get(status, "?"), status, msg))
        self.status = status


def library_path():
    """The in-tree library; NDTGPU_LIB names another build of it (A/B measurements of kernel variants)."""
    return os.environ.get("NDTGPU_LIB") or _SO


# per-source flags.  ndt_match.hip: the solver's pivoted LDL^T picks one of several statically indexed register swaps; with
# common-code sinking the optimiser merges those branches into ONE swap with computed indices, and the register array
# goes to the stack (the only private segment the matcher kernels would have).
# ndt_build_flat.hip: the same pass merges the two symmetric "replace run A / run B" branches into one that works through a
# POINTER to the run's scalar state, which then lives in scratch memory instead of scalar registers.
# ndt_build.hip: the same for the two runs of a lane in the 3D (SCAT) point loop -- their second moments were 112 B of
# scratch per lane and a memory round trip per point until round 4.
_SOURCE_FLAGS = {"ndt_match.hip": ["-mllvm", "-simplifycfg-sink-common=false"],
                 "ndt_build_flat.hip": ["-mllvm", "-simplifycfg-sink-common=false"],
                 "ndt_build.hip": ["-mllvm", "-simplifycfg-sink-common=false"]}


def build_library(force=False, verbose=False):
    """hipcc cross-compiles the kernels + C-ABI for gfx950 into the in-tree libndtgpu.so (one object per source, in
    parallel, then one link)."""
    csrc = os.path.join(_HERE, "csrc")
    so = library_path()
    srcs = [os.path.join(csrc, s) for s in _SOURCES]
    deps = srcs + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(_ROOT, "include", "ndtgpu.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(d) for d in deps):
        return so
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    extra = os.environ.get("NDTGPU_BUILD_FLAGS", "").split()      # experiments only (-DNDT_MATCH_PROF ...)
    # one object directory per output library: concurrent builds of variants (tools/build_variant.sh) do not share objects
    tag = os.path.splitext(os.path.basename(so))[0]
    objdir = os.path.join(_HERE, "build") if so == _SO else os.path.join(_HERE, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    # (spills never go to AGPRs: with AGPRs in use the matcher kernel would not keep 256 architectural VGPRs at two
    #  waves per SIMD)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical",
              "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", *extra]
    jobs = []
    for name, src in zip(_SOURCES, srcs):
        obj = os.path.join(objdir, name + ".o")
        cmd = common + _SOURCE_FLAGS.get(name, []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((subprocess.Popen(cmd), cmd, obj))
    objs, failed = [], None
    for proc, cmd, obj in jobs:                 # every compile job is reaped, also after a failure
        if proc.wait() != 0 and failed is None:
            failed = (proc.returncode, cmd)
        objs.append(obj)
    if failed:
        raise subprocess.CalledProcessError(*failed)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", so + ".tmp"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    os.replace(so + ".tmp", so)
    return so


class GridParams(C.Structure):
    _fields_ = [("res", C.c_double), ("centre", C.c_double * 3), ("size", C.c_double * 3), ("max_cells", C.c_uint32)]


class CellParams(C.Structure):
    _fields_ = [("n_min", C.c_int32), ("eval_factor", C.c_double)]


class FuseParams(C.Structure):
    _fields_ = [("maxz", C.c_double), ("sensor_noise", C.c_double), ("maxnumpoints", C.c_double),
                ("occupancy_limit", C.c_double), ("n_min", C.c_int32), ("eval_factor", C.c_double)]


class MatchParams(C.Structure):
    _fields_ = [("n_neighbours", C.c_int32), ("itr_max", C.c_int32), ("delta_score", C.c_double),
                ("step_control", C.c_int32), ("lfd1", C.c_double), ("lfd2", C.c_double), ("dof_mask", C.c_int32),
                ("use_initial_guess", C.c_int32)]


class MatchResult(C.Structure):
    _fields_ = [("converged", C.c_int32), ("iterations", C.c_int32), ("fevals", C.c_int32), ("exit_code", C.c_int32),
                ("score", C.c_double), ("n_source", C.c_int32), ("n_target", C.c_int32),
                ("cycles_eval", C.c_int64), ("cycles_solver", C.c_int64),
                ("pair_terms_g", C.c_int64), ("pair_terms_h", C.c_int64)]


RESULT_DTYPE = np.dtype([("converged", "<i4"), ("iterations", "<i4"), ("fevals", "<i4"), ("exit_code", "<i4"),
                         ("score", "<f8"), ("n_source", "<i4"), ("n_target", "<i4"), ("cycles_eval", "<i8"),
                         ("cycles_solver", "<i8"), ("pair_terms_g", "<i8"), ("pair_terms_h", "<i8")])
assert RESULT_DTYPE.itemsize == C.sizeof(MatchResult)

# every symbol include/ndtgpu.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = ["ndtgpu_version", "ndtgpu_last_error", "ndtgpu_device_count", "ndtgpu_default_cell_params",
           "ndtgpu_default_match_params", "ndtgpu_mapset_create", "ndtgpu_mapset_destroy", "ndtgpu_mapset_set_centre",
           "ndtgpu_mapset_info", "ndtgpu_mapset_build", "ndtgpu_mapset_build_host", "ndtgpu_mapset_num_cells",
           "ndtgpu_mapset_export_cells", "ndtgpu_mapset_set_cells", "ndtgpu_derivatives", "ndtgpu_match_batch",
           "ndtgpu_match_batch_device", "ndtgpu_match_d2d", "ndtgpu_kernel_name", "ndtgpu_profiling_enable",
           "ndtgpu_last_kernel_ms", "ndtgpu_mapset_counters", "ndtgpu_match_fusion_batch",
           "ndtgpu_mapset_enable_occupancy", "ndtgpu_default_fuse_params", "ndtgpu_mapset_add_cloud",
           "ndtgpu_mapset_add_cloud_host", "ndtgpu_mapset_clear", "ndtgpu_mapset_export_occupancy",
           "ndtgpu_overlap_score_batch", "ndtgpu_covariance_batch", "ndtgpu_mapset_discard_cells", "ndtgpu_mapset_import_occupancy",
           "ndtgpu_match_fusion_feat_batch", "ndtgpu_match_aborted", "ndtgpu_mapset_pack_bytes",
           "ndtgpu_mapset_pack_cells_device", "ndtgpu_mapset_unpack_cells_device", "ndtgpu_mapset_pack_bytes_sparse",
           "ndtgpu_mapset_occupied_cells_max", "ndtgpu_mapset_pack_cells_sparse_device", "ndtgpu_mapset_build_host_async",
           "ndtgpu_mapset_add_cloud_host_async", "ndtgpu_registrar_create", "ndtgpu_registrar_destroy",
           "ndtgpu_register_batch_device", "ndtgpu_registrar_wait_stream", "ndtgpu_registrar_sync",
           "ndtgpu_registrar_profiling", "ndtgpu_registrar_kernel_ms", "ndtgpu_registrar_mapset", "ndtgpu_register_batch_host",
           "ndtgpu_default_registrar_params", "ndtgpu_registrar_create_ex", "ndtgpu_registrar_get_info",
           "ndtgpu_default_fuser_params", "ndtgpu_fuser_prepare", "ndtgpu_fuser_bank_create", "ndtgpu_fuser_bank_destroy",
           "ndtgpu_fuser_bank_mapsets", "ndtgpu_fuser_initialize_batch", "ndtgpu_fuser_update_batch", "ndtgpu_fuser_poses",
           "ndtgpu_fuser_initialize_batch_host", "ndtgpu_fuser_update_batch_host", "ndtgpu_registrar_inject_abort"]

_lib = None


def lib():
    """Loads libndtgpu.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch bundles its own libamdhip64.so.7; load it first so that ONE HIP runtime serves both
        # torch (device memory, streams, torch.distributed) and libndtgpu.so (same soname -> shared).
        import torch  # noqa: F401
    except ImportError:
        pass
    so = library_path()
    if not os.path.exists(so):
        raise NdtGpuError(-2, "HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % so)
    L = C.CDLL(so)
    vp, dp, u32p, i32p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
    L.ndtgpu_version.restype = C.c_char_p
    L.ndtgpu_last_error.restype = C.c_char_p
    L.ndtgpu_kernel_name.restype = C.c_char_p
    L.ndtgpu_kernel_name.argtypes = [C.c_int]
    L.ndtgpu_device_count.restype = C.c_int
    L.ndtgpu_default_cell_params.argtypes = [C.POINTER(CellParams)]
    L.ndtgpu_default_match_params.argtypes = [C.POINTER(MatchParams)]
    L.ndtgpu_mapset_create.argtypes = [C.POINTER(GridParams), C.c_size_t, C.POINTER(vp)]
    L.ndtgpu_mapset_destroy.argtypes = [vp]
    L.ndtgpu_mapset_set_centre.argtypes = [vp, C.c_size_t, dp]
    L.ndtgpu_mapset_info.argtypes = [vp, C.POINTER(C.c_size_t), i32p, u32p]
    L.ndtgpu_mapset_build.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double,
                                      dp, C.POINTER(CellParams), vp]
    L.ndtgpu_mapset_build_host.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t,
                                           C.c_double, dp, C.POINTER(CellParams)]
    L.ndtgpu_mapset_build_host_async.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t,
                                                 C.c_double, dp, C.POINTER(CellParams), vp]
    L.ndtgpu_mapset_num_cells.argtypes = [vp, C.c_size_t, u32p]
    L.ndtgpu_mapset_export_cells.argtypes = [vp, C.c_size_t, dp, dp, i32p, u32p]
    L.ndtgpu_mapset_set_cells.argtypes = [vp, C.c_size_t, dp, dp, C.c_size_t]
    L.ndtgpu_derivatives.argtypes = [vp, C.c_size_t, dp, dp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double,
                                     dp, dp, dp]
    L.ndtgpu_match_batch.argtypes = [vp, u32p, vp, u32p, dp, C.c_size_t, C.POINTER(MatchParams), vp, vp]
    L.ndtgpu_match_fusion_batch.argtypes = [vp, u32p, vp, u32p, dp, dp, C.c_size_t, C.POINTER(MatchParams), C.c_int, vp, vp]
    L.ndtgpu_match_batch_device.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(MatchParams), vp, vp]
    L.ndtgpu_match_d2d.argtypes = [vp, C.c_size_t, vp, C.c_size_t, dp, C.POINTER(MatchParams), C.POINTER(MatchResult)]
    L.ndtgpu_profiling_enable.argtypes = [vp, C.c_int]
    L.ndtgpu_last_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.ndtgpu_mapset_counters.argtypes = [vp, C.c_size_t, u32p]
    L.ndtgpu_mapset_enable_occupancy.argtypes = [vp]
    L.ndtgpu_default_fuse_params.argtypes = [C.POINTER(FuseParams)]
    L.ndtgpu_mapset_add_cloud.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, dp,
                                          C.POINTER(FuseParams), vp]
    L.ndtgpu_mapset_add_cloud_host.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, dp,
                                               C.POINTER(FuseParams)]
    L.ndtgpu_mapset_add_cloud_host_async.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, dp,
                                                     C.POINTER(FuseParams), vp]
    L.ndtgpu_mapset_clear.argtypes = [vp, C.c_size_t, C.c_size_t]
    L.ndtgpu_mapset_export_occupancy.argtypes = [vp, C.c_size_t, C.POINTER(C.c_float)]
    L.ndtgpu_overlap_score_batch.argtypes = [vp, u32p, vp, u32p, dp, C.c_size_t, dp, C.POINTER(C.c_int64), vp]
    L.ndtgpu_mapset_import_occupancy.argtypes = [vp, C.c_size_t, C.POINTER(C.c_float)]
    L.ndtgpu_mapset_discard_cells.argtypes = [vp, C.c_size_t, C.POINTER(C.c_float), C.c_size_t]
    L.ndtgpu_covariance_batch.argtypes = [vp, u32p, vp, u32p, dp, C.c_size_t, C.POINTER(MatchParams), C.c_int, dp, i32p, vp]
    L.ndtgpu_match_aborted.argtypes = [vp, i32p]
    L.ndtgpu_mapset_pack_bytes.restype = C.c_size_t
    L.ndtgpu_mapset_pack_bytes.argtypes = [vp, C.c_uint32, C.c_int]
    L.ndtgpu_mapset_pack_cells_device.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_int, vp]
    L.ndtgpu_mapset_unpack_cells_device.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_int, vp]
    L.ndtgpu_mapset_pack_bytes_sparse.restype = C.c_size_t
    L.ndtgpu_mapset_pack_bytes_sparse.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.ndtgpu_mapset_occupied_cells_max.argtypes = [vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32), vp]
    L.ndtgpu_mapset_pack_cells_sparse_device.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp]
    L.ndtgpu_registrar_create.argtypes = [C.POINTER(GridParams), C.c_size_t, C.c_int, C.POINTER(vp)]
    L.ndtgpu_registrar_destroy.argtypes = [vp]
    L.ndtgpu_default_registrar_params.argtypes = [C.POINTER(RegistrarParams)]
    L.ndtgpu_default_registrar_params.restype = None
    L.ndtgpu_registrar_create_ex.argtypes = [C.POINTER(GridParams), C.POINTER(RegistrarParams), C.POINTER(vp)]
    L.ndtgpu_registrar_get_info.argtypes = [vp, C.POINTER(RegistrarInfo)]
    L.ndtgpu_registrar_inject_abort.argtypes = [vp]
    L.ndtgpu_default_fuser_params.argtypes = [C.POINTER(FuserParams)]
    L.ndtgpu_default_fuser_params.restype = None
    L.ndtgpu_fuser_prepare.argtypes = [C.POINTER(FuserParams), dp, dp, dp, C.POINTER(FuserPrepared)]
    L.ndtgpu_fuser_bank_create.argtypes = [C.POINTER(FuserParams), C.c_size_t, vp, C.POINTER(vp)]
    L.ndtgpu_fuser_bank_destroy.argtypes = [vp]
    L.ndtgpu_fuser_bank_mapsets.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.ndtgpu_fuser_initialize_batch.argtypes = [vp, C.c_size_t, C.c_size_t, dp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]
    L.ndtgpu_fuser_update_batch.argtypes = [vp, C.c_size_t, C.c_size_t, dp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, vp]
    L.ndtgpu_fuser_poses.argtypes = [vp, C.c_size_t, C.c_size_t, dp, vp]
    L.ndtgpu_register_batch_device.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(CellParams),
                                               vp, C.c_size_t, C.POINTER(MatchParams), vp, vp, C.POINTER(C.c_uint64)]
    L.ndtgpu_register_batch_host.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(CellParams),
                                             dp, C.c_size_t, C.POINTER(MatchParams), vp]
    L.ndtgpu_registrar_wait_stream.argtypes = [vp, C.c_uint64, vp]
    L.ndtgpu_registrar_sync.argtypes = [vp]
    L.ndtgpu_registrar_profiling.argtypes = [vp, C.c_int]
    L.ndtgpu_registrar_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), i32p]
    L.ndtgpu_registrar_mapset.argtypes = [vp, C.c_int, C.POINTER(vp)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise NdtGpuError(rc, lib().ndtgpu_last_error().decode())


def device_count():
    return lib().ndtgpu_device_count()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def match_params(**kw):
    p = MatchParams()
    lib().ndtgpu_default_match_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown match parameter %r" % k)
        setattr(p, k, v)
    return p


def _stream_ptr(stream):
    if stream is None:
        return None
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


class MapSet:
    """B x lslgeneric::NDTMap(new LazyGrid(res)) with one grid geometry, resident in HBM."""

    def __init__(self, res, centre, size_m, n_maps=1, max_cells=0):
        L = lib()
        gp = GridParams()
        gp.res = float(res)
        gp.centre[:] = [float(x) for x in centre]
        gp.size[:] = [float(x) for x in size_m]
        gp.max_cells = int(max_cells)
        h = C.c_void_p()
        _check(L.ndtgpu_mapset_create(C.byref(gp), int(n_maps), C.byref(h)))
        self.h = h
        self.n_maps = int(n_maps)
        self.res = float(res)

    def close(self):
        if getattr(self, "h", None):
            lib().ndtgpu_mapset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        n = C.c_size_t()
        cells = (C.c_int32 * 3)()
        cap = C.c_uint32()
        _check(lib().ndtgpu_mapset_info(self.h, C.byref(n), cells, C.byref(cap)))
        return dict(n_maps=n.value, cells_per_axis=list(cells), max_cells=cap.value)

    def set_centre(self, i, centre):
        c = _f64(centre)
        _check(lib().ndtgpu_mapset_set_centre(self.h, int(i), _dp(c)))

    def build(self, xyz, range_limit=-1.0, range_origins=None, first=0, n_min=3, eval_factor=1000.0, stream=None):
        """loadPointCloud + computeNDTCells for maps [first, first+B).  xyz: [B,N,3|4] float32,
        a torch CUDA tensor (device path, asynchronous) or a NumPy array (host path)."""
        cp = CellParams(int(n_min), float(eval_factor))
        ro = None
        is_torch = hasattr(xyz, "data_ptr")
        if xyz.ndim == 2:
            xyz = xyz[None]
        B, N, W = int(xyz.shape[0]), int(xyz.shape[1]), int(xyz.shape[2])
        assert W in (3, 4)
        if range_origins is not None:
            ro = _f64(range_origins).reshape(B, 3)
        rop = _dp(ro) if ro is not None else None
        if is_torch:
            import torch
            assert xyz.dtype == torch.float32 and xyz.is_cuda and xyz.is_contiguous()
            if stream is None:
                stream = torch.cuda.current_stream()
            _check(lib().ndtgpu_mapset_build(self.h, int(first), B, C.c_void_p(xyz.data_ptr()), N, 4 * W, 4 * W * N,
                                             float(range_limit), rop, C.byref(cp), _stream_ptr(stream)))
        else:
            a = np.ascontiguousarray(xyz, dtype=np.float32)
            if stream is not None:      # asynchronous host form: returns when `a` has been read, the maps are ready when `stream` is
                _check(lib().ndtgpu_mapset_build_host_async(self.h, int(first), B, C.c_void_p(a.ctypes.data), N, 4 * W,
                                                            4 * W * N, float(range_limit), rop, C.byref(cp), _stream_ptr(stream)))
            else:
                _check(lib().ndtgpu_mapset_build_host(self.h, int(first), B, C.c_void_p(a.ctypes.data), N, 4 * W,
                                                      4 * W * N, float(range_limit), rop, C.byref(cp)))

    def enable_occupancy(self):
        """NDTMap::initialize: every cell exists and carries an occupancy; needed by add_cloud / overlap_score."""
        _check(lib().ndtgpu_mapset_enable_occupancy(self.h))

    def add_cloud(self, xyz, origins, first=0, stream=None, **params):
        """NDTMap::addPointCloud(origin, cloud, ., maxz, sensor_noise) + computeNDTCells(SAMPLE_VARIANCE, maxnumpoints,
        occupancy_limit, ..) for maps [first, first+B): xyz [B,N,3|4] float32 (torch CUDA tensor: asynchronous; NumPy:
        host path), origins [B,3] sensor positions in the map frame."""
        fp = FuseParams()
        lib().ndtgpu_default_fuse_params(C.byref(fp))
        for k, v in params.items():
            if not hasattr(fp, k):
                raise TypeError("unknown fuse parameter %r" % k)
            setattr(fp, k, v)
        is_torch = hasattr(xyz, "data_ptr")
        if xyz.ndim == 2:
            xyz = xyz[None]
        B, N, W = int(xyz.shape[0]), int(xyz.shape[1]), int(xyz.shape[2])
        assert W in (3, 4)
        org = _f64(origins).reshape(B, 3)
        if is_torch:
            import torch
            assert xyz.dtype == torch.float32 and xyz.is_cuda and xyz.is_contiguous()
            if stream is None:
                stream = torch.cuda.current_stream()
            _check(lib().ndtgpu_mapset_add_cloud(self.h, int(first), B, C.c_void_p(xyz.data_ptr()), N, 4 * W, 4 * W * N,
                                                 _dp(org), C.byref(fp), _stream_ptr(stream)))
        else:
            a = np.ascontiguousarray(xyz, dtype=np.float32)
            _check(lib().ndtgpu_mapset_add_cloud_host(self.h, int(first), B, C.c_void_p(a.ctypes.data), N, 4 * W, 4 * W * N,
                                                      _dp(org), C.byref(fp)))

    def discard_cells(self, i, xyz):
        """ndt_feature::discardCell for every point: the cells that hold them lose their Gaussian."""
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        _check(lib().ndtgpu_mapset_discard_cells(self.h, int(i), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0]))

    def clear(self, first=0, count=None):
        _check(lib().ndtgpu_mapset_clear(self.h, int(first), int(self.n_maps - first if count is None else count)))

    def occupancy(self, i=0):
        """NDTCell::occ of every cell of map i, shape (sx, sy, sz)."""
        shape = tuple(self.info()["cells_per_axis"])
        out = np.zeros(shape, dtype=np.float32)
        _check(lib().ndtgpu_mapset_export_occupancy(self.h, int(i), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def pack_bytes(self, cells_cap, with_occupancy=False, occ_cap=None):
        """Bytes of one exchange record.  occ_cap: the sparse form of the occupancy block (room for that many cells with a
        reading, ndtgpu_mapset_pack_bytes_sparse) instead of one float per slot."""
        if occ_cap is not None:
            return int(lib().ndtgpu_mapset_pack_bytes_sparse(self.h, int(cells_cap), int(occ_cap)))
        return int(lib().ndtgpu_mapset_pack_bytes(self.h, int(cells_cap), int(bool(with_occupancy))))

    def occupied_cells_max(self, first=0, count=None, stream=None):
        """The largest number of cells with an occupancy reading over maps [first, first+count): what the occ_cap of a
        sparse exchange record must hold (ndtgpu_mapset_occupied_cells_max; synchronises the stream)."""
        import torch
        count = self.n_maps - first if count is None else count
        if stream is None:
            stream = torch.cuda.current_stream()
        out = C.c_uint32()
        _check(lib().ndtgpu_mapset_occupied_cells_max(self.h, int(first), int(count), C.byref(out), _stream_ptr(stream)))
        return int(out.value)

    def pack_cells(self, buf, first=0, count=None, cells_cap=None, with_occupancy=False, stream=None, occ_cap=None):
        """Exchange records of maps [first, first+count) into the torch CUDA uint8 tensor buf [count, stride]
        (ndtgpu_mapset_pack_cells_device; asynchronous).  occ_cap: occupancies as (slot, value) pairs of the cells with a
        reading (ndtgpu_mapset_pack_cells_sparse_device)."""
        import torch
        count = self.n_maps - first if count is None else count
        cap = self.info()["max_cells"] if cells_cap is None else cells_cap
        assert buf.dtype == torch.uint8 and buf.is_cuda and buf.is_contiguous() and buf.shape[0] >= count
        if stream is None:
            stream = torch.cuda.current_stream()
        if occ_cap is not None:
            _check(lib().ndtgpu_mapset_pack_cells_sparse_device(self.h, int(first), int(count), C.c_void_p(buf.data_ptr()),
                                                                int(buf.stride(0)), int(cap), int(occ_cap), _stream_ptr(stream)))
            return
        _check(lib().ndtgpu_mapset_pack_cells_device(self.h, int(first), int(count), C.c_void_p(buf.data_ptr()), int(buf.stride(0)),
                                                     int(cap), int(bool(with_occupancy)), _stream_ptr(stream)))

    def unpack_cells(self, buf, first=0, count=None, with_occupancy=False, stream=None):
        """Installs exchange records (ndtgpu_mapset_unpack_cells_device; asynchronous)."""
        import torch
        count = buf.shape[0] if count is None else count
        assert buf.dtype == torch.uint8 and buf.is_cuda and buf.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream()
        _check(lib().ndtgpu_mapset_unpack_cells_device(self.h, int(first), int(count), C.c_void_p(buf.data_ptr()), int(buf.stride(0)),
                                                       int(bool(with_occupancy)), _stream_ptr(stream)))

    def profiling(self, on=True):
        _check(lib().ndtgpu_profiling_enable(self.h, int(bool(on))))

    def last_kernel_ms(self, which):
        """HIP-event duration of the most recent build (0) / match (1) kernel launched on this set."""
        ms = C.c_float()
        _check(lib().ndtgpu_last_kernel_ms(self.h, int(which), C.byref(ms)))
        return ms.value

    def counters(self, i=0):
        out = (C.c_uint32 * 8)()
        _check(lib().ndtgpu_mapset_counters(self.h, int(i), out))
        return dict(n_alloc=out[0], n_cells=out[1], overflow=out[2], n_dropped=out[3], cyc=list(out[4:8]))

    def num_cells_all(self):
        """n_cells of every map (one D2H of the counters; synchronises)."""
        return np.array([self.num_cells(i) for i in range(self.n_maps)], dtype=np.int64)

    def num_cells(self, i=0):
        n = C.c_uint32()
        _check(lib().ndtgpu_mapset_num_cells(self.h, int(i), C.byref(n)))
        return n.value

    def export_cells(self, i=0):
        n = self.num_cells(i)
        mean = np.zeros((n, 3))
        cov = np.zeros((n, 3, 3))
        idx = np.zeros((n, 3), dtype=np.int32)
        npts = np.zeros(n, dtype=np.uint32)
        _check(lib().ndtgpu_mapset_export_cells(self.h, int(i), _dp(mean), _dp(cov),
                                                idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                                npts.ctypes.data_as(C.POINTER(C.c_uint32))))
        return mean, cov, idx, npts

    def set_cells(self, i, mean, cov):
        mean, cov = _f64(mean).reshape(-1, 3), _f64(cov).reshape(-1, 3, 3)
        _check(lib().ndtgpu_mapset_set_cells(self.h, int(i), _dp(mean), _dp(cov), mean.shape[0]))


def derivatives(target, tmap, src_mean, src_cov, n_neighbours=2, compute_hessian=True, lfd1=1.0, lfd2=0.05):
    """NDTMatcherD2D::derivativesNDT on the device (source cells already in the target frame)."""
    src_mean, src_cov = _f64(src_mean).reshape(-1, 3), _f64(src_cov).reshape(-1, 3, 3)
    s = C.c_double()
    g = np.zeros(6)
    H = np.zeros((6, 6))
    _check(lib().ndtgpu_derivatives(target.h, int(tmap), _dp(src_mean), _dp(src_cov), src_mean.shape[0],
                                    int(n_neighbours), int(bool(compute_hessian)), lfd1, lfd2, C.byref(s), _dp(g), _dp(H)))
    return s.value, g, H


def match_batch(target_set, target_idx, source_set, source_idx, T, stream=None, **params):
    """NDTFeatureGraph::updateLinksUsingNDTRegistration's loop: match(target, source, T, true) for every
    pair.  T: [n,4,4] (math convention); returns (T_out [n,4,4], results structured array)."""
    ti = np.ascontiguousarray(target_idx, dtype=np.uint32)
    si = np.ascontiguousarray(source_idx, dtype=np.uint32)
    n = ti.shape[0]
    Tc = np.ascontiguousarray(np.transpose(np.asarray(T, dtype=np.float64).reshape(n, 4, 4), (0, 2, 1))).copy()
    res = np.zeros(n, dtype=RESULT_DTYPE)
    p = match_params(**params)
    _check(lib().ndtgpu_match_batch(target_set.h, ti.ctypes.data_as(C.POINTER(C.c_uint32)), source_set.h,
                                    si.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(Tc), n, C.byref(p),
                                    C.c_void_p(res.ctypes.data), _stream_ptr(stream)))
    return np.transpose(Tc, (0, 2, 1)).copy(), res


def match_fusion_batch(target_set, target_idx, source_set, source_idx, T, Tcov, use_soft_constraints=True, stream=None,
                       tikhonov=False, **params):
    """ndt_feature::matchFusion (NDT term + odometry soft constraint and / or Tikhonov regularisation) for every pair.
    Tcov: [n,6,6]."""
    ti = np.ascontiguousarray(target_idx, dtype=np.uint32)
    si = np.ascontiguousarray(source_idx, dtype=np.uint32)
    n = ti.shape[0]
    Tc = np.ascontiguousarray(np.transpose(np.asarray(T, dtype=np.float64).reshape(n, 4, 4), (0, 2, 1))).copy()
    cov = np.ascontiguousarray(np.asarray(Tcov, dtype=np.float64).reshape(n, 36))
    res = np.zeros(n, dtype=RESULT_DTYPE)
    p = match_params(**params)
    _check(lib().ndtgpu_match_fusion_batch(target_set.h, ti.ctypes.data_as(C.POINTER(C.c_uint32)), source_set.h,
                                           si.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(Tc), _dp(cov), n, C.byref(p),
                                           int(bool(use_soft_constraints)) | (2 if tikhonov else 0), C.c_void_p(res.ctypes.data),
                                           _stream_ptr(stream)))
    return np.transpose(Tc, (0, 2, 1)).copy(), res


class FeatPairs(C.Structure):
    _fields_ = [("offsets", C.POINTER(C.c_uint32)), ("src_mean", C.POINTER(C.c_double)), ("src_cov", C.POINTER(C.c_double)),
                ("tgt_mean", C.POINTER(C.c_double)), ("tgt_cov", C.POINTER(C.c_double))]


def match_fusion_feat_batch(target_set, target_idx, source_set, source_idx, T, Tcov, feat, use_soft_constraints=True,
                            tikhonov=False, step_control_fusion=False, stream=None, **params):
    """ndt_feature::matchFusion with feature / odometry-cell maps.  feat: one entry per pair, each a tuple
    (src_mean [k,3], src_cov [k,6], tgt_mean [k,3], tgt_cov [k,6]) of k <= 64 corresponding cells (k may be 0)."""
    ti = np.ascontiguousarray(target_idx, dtype=np.uint32)
    si = np.ascontiguousarray(source_idx, dtype=np.uint32)
    n = ti.shape[0]
    Tc = np.ascontiguousarray(np.transpose(np.asarray(T, dtype=np.float64).reshape(n, 4, 4), (0, 2, 1))).copy()
    cov = np.ascontiguousarray(np.asarray(Tcov, dtype=np.float64).reshape(n, 36))
    off = np.zeros(n + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(f[0]) for f in feat])
    cat = lambda j, w: np.ascontiguousarray(np.concatenate([np.asarray(f[j], dtype=np.float64).reshape(-1, w) for f in feat]
                                                           + [np.zeros((1, w))]))
    sm, sc, tm, tc = cat(0, 3), cat(1, 6), cat(2, 3), cat(3, 6)
    fp = FeatPairs(off.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(sm), _dp(sc), _dp(tm), _dp(tc))
    res = np.zeros(n, dtype=RESULT_DTYPE)
    p = match_params(**params)
    L = lib()
    L.ndtgpu_match_fusion_feat_batch.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32),
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(FeatPairs), C.c_size_t,
                                                 C.POINTER(MatchParams), C.c_int, C.c_void_p, C.c_void_p]
    flags = int(bool(use_soft_constraints)) | (2 if tikhonov else 0) | (4 if step_control_fusion else 0)
    _check(L.ndtgpu_match_fusion_feat_batch(target_set.h, ti.ctypes.data_as(C.POINTER(C.c_uint32)), source_set.h,
                                            si.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(Tc), _dp(cov), C.byref(fp), n,
                                            C.byref(p), flags, C.c_void_p(res.ctypes.data), _stream_ptr(stream)))
    return np.transpose(Tc, (0, 2, 1)).copy(), res


def match_aborted(target_set):
    """True when the last persistent matcher launch on this target set gave up (waits for that launch)."""
    a = C.c_int32()
    _check(lib().ndtgpu_match_aborted(target_set.h, C.byref(a)))
    return bool(a.value)


def match_batch_device(target_set, tidx_dev, source_set, sidx_dev, T16_dev, results_dev, n_pairs, stream=None, **params):
    """Asynchronous device-resident variant (torch CUDA tensors: uint32/int32 idx, float64 [n,16]
    column-major poses, uint8 [n,64] results)."""
    p = match_params(**params)
    _check(lib().ndtgpu_match_batch_device(target_set.h, C.c_void_p(tidx_dev.data_ptr()), source_set.h,
                                           C.c_void_p(sidx_dev.data_ptr()), C.c_void_p(T16_dev.data_ptr()), int(n_pairs),
                                           C.byref(p), C.c_void_p(results_dev.data_ptr()), _stream_ptr(stream)))


class _BorrowedMapSet(MapSet):
    """A map set owned by somebody else (a registrar): same methods, never destroyed from here."""

    def __init__(self, h, n_maps, res):
        self.h, self.n_maps, self.res = h, int(n_maps), float(res)

    def close(self):
        self.h = None


class RegistrarParams(C.Structure):
    _fields_ = [("pairs_per_batch", C.c_size_t), ("depth", C.c_int32), ("matcher_form", C.c_int32), ("matcher_groups", C.c_uint32),
                ("build_streams", C.c_int32), ("linger_us", C.c_uint32), ("recalibrate_pct", C.c_int32), ("matcher_slots", C.c_int32)]


class RegistrarInfo(C.Structure):
    _fields_ = [("matcher_form", C.c_int32), ("matcher_groups", C.c_uint32), ("build_streams", C.c_int32), ("calibrations", C.c_int32),
                ("submitted", C.c_uint64), ("cells_per_map", C.c_double), ("matcher_slots", C.c_int32), ("resident_groups", C.c_int32)]


MATCHER_AUTO, MATCHER_PER_BATCH, MATCHER_STREAM_FED = 0, 1, 2


class Registrar:
    """ndtgpu_registrar: scans in, poses out -- grid builds + D2D matcher of batches of scan pairs as ONE asynchronous call,
    pipelined over the library's own streams (include/ndtgpu.h).  Keyword arguments beyond pairs_per_batch / depth are the
    fields of ndtgpu_registrar_params (matcher_form, matcher_groups, build_streams, linger_us, recalibrate_pct)."""

    def __init__(self, res, centre, size_m, pairs_per_batch=1024, depth=8, max_cells=0, **fields):
        gp = GridParams()
        gp.res = float(res)
        gp.centre[:] = [float(x) for x in centre]
        gp.size[:] = [float(x) for x in size_m]
        gp.max_cells = int(max_cells)
        rp = RegistrarParams()
        lib().ndtgpu_default_registrar_params(C.byref(rp))
        rp.pairs_per_batch, rp.depth = int(pairs_per_batch), int(depth)
        for k, v in fields.items():
            if k not in dict(RegistrarParams._fields_):
                raise TypeError("Registrar: no parameter %r" % k)
            setattr(rp, k, int(v))
        h = C.c_void_p()
        _check(lib().ndtgpu_registrar_create_ex(C.byref(gp), C.byref(rp), C.byref(h)))
        self.h, self.depth, self.per, self.res = h, int(depth), int(pairs_per_batch), float(res)

    def inject_abort(self):
        """test aid: the stream-fed matcher's give-up word (include/ndtgpu.h)"""
        _check(lib().ndtgpu_registrar_inject_abort(self.h))

    def info(self):
        i = RegistrarInfo()
        _check(lib().ndtgpu_registrar_get_info(self.h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in RegistrarInfo._fields_}

    def close(self):
        if getattr(self, "h", None):
            lib().ndtgpu_registrar_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, targets, sources, T16_dev, results_dev, range_limit=-1.0, n_min=3, eval_factor=1000.0, stream=None, **params):
        """targets / sources: torch CUDA float32 tensors [n, N, 3 or 4] (contiguous in the last two dims); T16_dev float64
        [n, 16] column-major (in: initial guess, out: pose); results_dev uint8 [n, 64].  Asynchronous; returns the call's
        ticket (wait_stream / sync)."""
        n, npts = int(targets.shape[0]), int(targets.shape[1])
        assert tuple(sources.shape) == tuple(targets.shape) and targets.stride(1) == targets.shape[2] and targets.stride(2) == 1
        assert sources.stride(0) == targets.stride(0) and sources.stride(1) == targets.stride(1) and sources.stride(2) == 1
        cp = CellParams(int(n_min), float(eval_factor))
        p = match_params(**params)
        t = C.c_uint64()
        _check(lib().ndtgpu_register_batch_device(self.h, C.c_void_p(targets.data_ptr()), C.c_void_p(sources.data_ptr()), npts,
                                                  4 * int(targets.shape[2]), 4 * int(targets.stride(0)), float(range_limit),
                                                  C.byref(cp), C.c_void_p(T16_dev.data_ptr()), n, C.byref(p),
                                                  C.c_void_p(results_dev.data_ptr()), _stream_ptr(stream), C.byref(t)))
        return int(t.value)

    def register_host(self, targets, sources, T, range_limit=-1.0, n_min=3, eval_factor=1000.0, **params):
        """Host form: targets / sources NumPy float32 [n, N, 3 or 4], T [n, 4, 4] initial guesses -> (T [n, 4, 4], results)."""
        tg = np.ascontiguousarray(targets, dtype=np.float32)
        sc = np.ascontiguousarray(sources, dtype=np.float32)
        n, npts, w = tg.shape
        assert sc.shape == tg.shape and w in (3, 4)
        Tc = np.ascontiguousarray(np.transpose(np.asarray(T, dtype=np.float64).reshape(n, 4, 4), (0, 2, 1))).copy()
        res = np.zeros(n, dtype=RESULT_DTYPE)
        cp = CellParams(int(n_min), float(eval_factor))
        p = match_params(**params)
        _check(lib().ndtgpu_register_batch_host(self.h, C.c_void_p(tg.ctypes.data), C.c_void_p(sc.ctypes.data), npts, 4 * w, 4 * w * npts,
                                                float(range_limit), C.byref(cp), _dp(Tc), n, C.byref(p), C.c_void_p(res.ctypes.data)))
        return np.transpose(Tc.reshape(n, 4, 4), (0, 2, 1)).copy(), res

    def wait_stream(self, stream=None, ticket=0):
        """`stream` waits for the call `ticket` names (0: for everything submitted so far); the host does not wait."""
        _check(lib().ndtgpu_registrar_wait_stream(self.h, int(ticket), _stream_ptr(stream)))

    def sync(self):
        _check(lib().ndtgpu_registrar_sync(self.h))

    def profiling(self, on=True):
        _check(lib().ndtgpu_registrar_profiling(self.h, 1 if on else 0))

    def kernel_ms(self):
        """-> (mean build ms, mean matcher ms, profiled sub-batches) since the last call; forgets them."""
        ms = (C.c_float * 2)()
        n = C.c_int32()
        _check(lib().ndtgpu_registrar_kernel_ms(self.h, ms, C.byref(n)))
        return float(ms[0]), float(ms[1]), int(n.value)

    def mapset(self, slot):
        h = C.c_void_p()
        _check(lib().ndtgpu_registrar_mapset(self.h, int(slot), C.byref(h)))
        return _BorrowedMapSet(h, 2 * self.per, self.res)


class FuserParams(C.Structure):
    _fields_ = [("resolution", C.c_double), ("map_size_x", C.c_double), ("map_size_y", C.c_double), ("map_size_z", C.c_double),
                ("sensor_range", C.c_double), ("max_translation_norm", C.c_double), ("max_rotation_norm", C.c_double),
                ("check_consistency", C.c_int32), ("fuse_incomplete", C.c_int32), ("use_odom", C.c_int32), ("neighbours", C.c_int32),
                ("stepcontrol", C.c_int32), ("itr_max", C.c_int32), ("delta_score", C.c_double), ("force_odom_as_est", C.c_int32),
                ("fusion2d", C.c_int32), ("all_matches_valid", C.c_int32), ("use_soft_constraints", C.c_int32), ("compute_cov", C.c_int32),
                ("step_control_fusion", C.c_int32), ("use_tikhonov", C.c_int32), ("covariance_mode", C.c_int32),
                ("discard_cells", C.c_int32), ("pad_", C.c_int32),
                ("motion_Cd", C.c_double), ("motion_Ct", C.c_double), ("motion_Dd", C.c_double), ("motion_Dt", C.c_double),
                ("motion_Td", C.c_double), ("motion_Tt", C.c_double), ("sensor_pose", C.c_double * 16), ("max_cells", C.c_uint32)]


class FuserPrepared(C.Structure):
    _fields_ = [("Tscan", C.c_double * 16), ("scan_centre", C.c_double * 3), ("range_origin", C.c_double * 3), ("Tcov", C.c_double * 36),
                ("odom_cov", C.c_double * 9), ("feat_src_mean", C.c_double * 3), ("feat_tgt_mean", C.c_double * 3),
                ("feat_cov_rotated", C.c_double * 6), ("feat_cov_plain", C.c_double * 6)]


FUSER_RESULT_DTYPE = np.dtype([("Tnow", "<f8", (16,)), ("Tmotion_est", "<f8", (16,)), ("spose", "<f8", (16,)), ("match", RESULT_DTYPE),
                               ("match_ok", "<i4"), ("registration_failure", "<i4"), ("cov_singular", "<i4"), ("pad_", "<i4"),
                               ("posecov_mean", "<f8", (3,)), ("posecov", "<f8", (9,))])


def fuser_params(**fields):
    """ndtgpu_default_fuser_params with fields overridden; sensor_pose: 4x4 (row-major NumPy) or 16 column-major numbers."""
    p = FuserParams()
    lib().ndtgpu_default_fuser_params(C.byref(p))
    for k, v in fields.items():
        if k == "sensor_pose":
            a = np.asarray(v, dtype=np.float64)
            p.sensor_pose[:] = list(a.T.reshape(16) if a.shape == (4, 4) else a.reshape(16))
        elif k not in dict(FuserParams._fields_):
            raise TypeError("fuser_params: no field %r" % k)
        else:
            setattr(p, k, v)
    return p


def fuser_prepare(params, Tnow, Tmotion, node_centre):
    """ndtgpu_fuser_prepare: what the host derives from (pose, odometry increment).  Tnow / Tmotion: 4x4 row-major NumPy."""
    out = FuserPrepared()
    tn = np.ascontiguousarray(np.asarray(Tnow, dtype=np.float64).T).reshape(16).copy()
    tm = np.ascontiguousarray(np.asarray(Tmotion, dtype=np.float64).T).reshape(16).copy()
    nc = _f64(node_centre).reshape(3)
    _check(lib().ndtgpu_fuser_prepare(C.byref(params), _dp(tn), _dp(tm), _dp(nc), C.byref(out)))
    return {k: np.array(getattr(out, k)) for k, _ in FuserPrepared._fields_}


class FuserBank:
    """ndtgpu_fuser_bank: NDTFeatureFuserHMT::initialize / update for a batch of independent fusers, one asynchronous call
    each (include/ndtgpu.h).  Poses are 4x4 row-major NumPy arrays on this side."""

    def __init__(self, params, n_fusers, node_maps=None):
        h = C.c_void_p()
        _check(lib().ndtgpu_fuser_bank_create(C.byref(params), int(n_fusers), node_maps.h if node_maps is not None else None, C.byref(h)))
        self.h, self.n, self.params = h, int(n_fusers), params
        self._node_maps = node_maps       # (kept alive)

    def close(self):
        if getattr(self, "h", None):
            lib().ndtgpu_fuser_bank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def mapsets(self):
        a, b = C.c_void_p(), C.c_void_p()
        _check(lib().ndtgpu_fuser_bank_mapsets(self.h, C.byref(a), C.byref(b)))
        return _BorrowedMapSet(a, self.n, self.params.resolution), _BorrowedMapSet(b, self.n, self.params.resolution)

    @staticmethod
    def _poses16(T, n):
        return np.ascontiguousarray(np.transpose(np.asarray(T, dtype=np.float64).reshape(n, 4, 4), (0, 2, 1))).reshape(n, 16).copy()

    def _cloud_args(self, xyz):
        import torch
        assert xyz.dtype == torch.float32 and xyz.is_cuda and xyz.stride(2) == 1 and xyz.stride(1) == xyz.shape[2]
        return C.c_void_p(xyz.data_ptr()), int(xyz.shape[1]), 4 * int(xyz.shape[2]), 4 * int(xyz.stride(0))

    def initialize(self, init_poses, xyz, first=0, stream=None):
        n = int(xyz.shape[0])
        T = self._poses16(init_poses, n)
        ptr, npts, stride, mstride = self._cloud_args(xyz)
        _check(lib().ndtgpu_fuser_initialize_batch(self.h, int(first), n, _dp(T), ptr, npts, stride, mstride, _stream_ptr(stream)))

    def update(self, Tmotion, xyz, first=0, update_ndt_map=True, stream=None):
        n = int(xyz.shape[0])
        T = self._poses16(Tmotion, n)
        ptr, npts, stride, mstride = self._cloud_args(xyz)
        _check(lib().ndtgpu_fuser_update_batch(self.h, int(first), n, _dp(T), ptr, npts, stride, mstride, 1 if update_ndt_map else 0,
                                               _stream_ptr(stream)))

    def poses(self, first=0, count=None):
        """-> (Tnow [n,4,4] row-major, results of the last update call: FUSER_RESULT_DTYPE [n]); waits for the bank"""
        n = self.n - first if count is None else int(count)
        T = np.zeros((n, 16))
        res = np.zeros(n, dtype=FUSER_RESULT_DTYPE)
        _check(lib().ndtgpu_fuser_poses(self.h, int(first), n, _dp(T), C.c_void_p(res.ctypes.data)))
        return np.transpose(T.reshape(n, 4, 4), (0, 2, 1)).copy(), res


def match_d2d(target_set, tmap, source_set, smap, T, **params):
    To, res = match_batch(target_set, [tmap], source_set, [smap], np.asarray(T)[None], **params)
    return To[0], res[0]


def overlap_score(ref_set, ref_idx, mov_set, mov_idx, T, stream=None):
    """ndt_feature::overlapNDTOccupancyScore(ref, mov, T) for every link -> (scores [n], nb_sum [n])."""
    ri = np.ascontiguousarray(ref_idx, dtype=np.uint32)
    mi = np.ascontiguousarray(mov_idx, dtype=np.uint32)
    n = ri.shape[0]
    Tc = np.ascontiguousarray(np.transpose(np.asarray(T, dtype=np.float64).reshape(n, 4, 4), (0, 2, 1))).copy()
    score = np.zeros(n)
    nb = np.zeros(n, dtype=np.int64)
    _check(lib().ndtgpu_overlap_score_batch(ref_set.h, ri.ctypes.data_as(C.POINTER(C.c_uint32)), mov_set.h,
                                            mi.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(Tc), n, _dp(score),
                                            nb.ctypes.data_as(C.POINTER(C.c_int64)), _stream_ptr(stream)))
    return score, nb


def covariance(target_set, target_idx, source_set, source_idx, T, mode=0, stream=None, **params):
    """NDTMatcherD2D::covariance(target, source, T, cov) for every link -> (cov [n,6,6], singular [n])."""
    ti = np.ascontiguousarray(target_idx, dtype=np.uint32)
    si = np.ascontiguousarray(source_idx, dtype=np.uint32)
    n = ti.shape[0]
    Tc = np.ascontiguousarray(np.transpose(np.asarray(T, dtype=np.float64).reshape(n, 4, 4), (0, 2, 1))).copy()
    cov = np.zeros((n, 6, 6))
    sing = np.zeros(n, dtype=np.int32)
    p = match_params(**params)
    _check(lib().ndtgpu_covariance_batch(target_set.h, ti.ctypes.data_as(C.POINTER(C.c_uint32)), source_set.h,
                                         si.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(Tc), n, C.byref(p), int(mode), _dp(cov),
                                         sing.ctypes.data_as(C.POINTER(C.c_int32)), _stream_ptr(stream)))
    return cov, sing
