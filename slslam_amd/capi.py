"""ctypes binding of the C ABI in include/slslam_hip.h (libslslam_hip.so, built in-tree).

The Python layer is plumbing for tests, benches and multi-GPU fan-out; the product is the HIP
library.  There is no CPU fallback here: if the library is missing or no HIP device is usable
the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libslslam_hip.so")

OK = 0
STATUS = {0: "ok", 1: "invalid argument", 2: "no usable HIP device", 3: "HIP runtime error",
          4: "unsupported problem shape", 5: "invalid call sequence", 6: "host allocation failed"}
TERMINATION = {0: "NO_CONVERGENCE", 1: "GRADIENT_TOLERANCE", 2: "FUNCTION_TOLERANCE",
               3: "PARAMETER_TOLERANCE", 4: "NUMERICAL_FAILURE", 5: "MIN_RADIUS"}
KERNEL_FAMILIES = ["linearise_schur", "reduced_solve", "backsub", "line_trig", "candidate_cost",
                   "lm_update", "init_linearise", "unused"]


class SlslamError(RuntimeError):
    def __init__(self, status, where):
        super().__init__("%s: %s (status %d)" % (where, STATUS.get(status, "?"), status))
        self.status = status


class SolverOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("huber_delta", C.c_double), ("baseline", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("max_num_consecutive_invalid_steps", C.c_int), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("jacobi_scaling", C.c_int), ("use_graph", C.c_int), ("chunks_per_window", C.c_int),
                ("reuse_elimination", C.c_int), ("po_factor_fp32", C.c_int), ("po_dense_factor", C.c_int), ("lba_fused_motion_only", C.c_int),
                ("lba_elimination", C.c_int), ("lba_keep_jacobian", C.c_int), ("refill_headroom_percent", C.c_int),
                ("host_threads", C.c_int), ("reproducible", C.c_int), ("lba_precision", C.c_int), ("device_build", C.c_int)]


class Summary(C.Structure):
    _fields_ = [("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
                ("termination_type", C.c_int), ("num_free_parameters", C.c_int),
                ("num_residual_blocks", C.c_int)]


class Iteration(C.Structure):
    _fields_ = [("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int),
                ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double),
                ("trust_region_radius", C.c_double), ("model_cost_change", C.c_double)]


class LBAWindow(C.Structure):
    _fields_ = [("num_cameras", C.c_int), ("num_lines", C.c_int), ("num_observations", C.c_int),
                ("camera_index", C.POINTER(C.c_int)), ("line_index", C.POINTER(C.c_int)),
                ("fixed_index", C.POINTER(C.c_int)), ("observations", C.POINTER(C.c_double)),
                ("parameters", C.POINTER(C.c_double))]


class RansacTrials(C.Structure):
    _fields_ = [("num_trials", C.c_int), ("sample_size", C.c_int), ("num_lines", C.c_int), ("samples", C.POINTER(C.c_int)),
                ("observations0", C.POINTER(C.c_double)), ("observations1", C.POINTER(C.c_double))]


class RansacFrame(C.Structure):
    _fields_ = [("num_hypotheses", C.c_int), ("num_lines", C.c_int), ("poses", C.POINTER(C.c_double)),
                ("observations", C.POINTER(C.c_double)), ("lines", C.POINTER(C.c_double))]


class POGraph(C.Structure):
    _fields_ = [("num_poses", C.c_int), ("num_edges", C.c_int),
                ("pose_index_1", C.POINTER(C.c_int)), ("pose_index_2", C.POINTER(C.c_int)),
                ("constraints", C.POINTER(C.c_double)), ("parameters", C.POINTER(C.c_double))]


# every symbol include/slslam_hip.h declares (checked by the CPU test-suite)
EXPORTS = [
    "slslam_default_options", "slslam_lba_solve", "slslam_lba_batch_create", "slslam_lba_batch_destroy",
    "slslam_lba_batch_add", "slslam_lba_batch_finalize", "slslam_lba_batch_solve", "slslam_lba_batch_reset",
    "slslam_lba_batch_download", "slslam_lba_batch_download_async", "slslam_lba_batch_wait", "slslam_lba_batch_refill",
    "slslam_lba_stream_create", "slslam_lba_stream_destroy", "slslam_lba_stream_submit", "slslam_lba_stream_collect", "slslam_lba_stream_stats",
    "slslam_lba_stream_build_stats", "slslam_lba_stream_batch", "slslam_lba_stream_submit_packed", "slslam_pinned_alloc", "slslam_pinned_free", "slslam_pinned_register", "slslam_pinned_unregister",
    "slslam_pinned_contains", "slslam_pack_indices", "slslam_debug_device_pack", "slslam_debug_device_pack_timed",
    "slslam_lba_batch_get_parameters", "slslam_lba_batch_get_summary",
    "slslam_lba_batch_get_trace", "slslam_lba_batch_export_device", "slslam_lba_batch_counts", "slslam_lba_batch_window_chunks", "slslam_lba_batch_path", "slslam_lba_batch_elimination",
    "slslam_lba_batch_iterations", "slslam_lba_batch_set_profiling", "slslam_lba_batch_kernel_times", "slslam_lba_batch_linearise",
    "slslam_po_solve", "slslam_po_structure", "slslam_po_structure_level1", "slslam_po_set_profiling", "slslam_po_last_timing", "slslam_debug_phase_cycles", "slslam_debug_read_cycles", "slslam_ransac_score", "slslam_ransac_generate", "slslam_ransac_motion", "slslam_ransac_motion_batch", "slslam_device_count", "slslam_release_cached_memory", "slslam_version", "slslam_status_string",
]

_lib = None


def lib():
    """Load the in-tree HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # SLSLAM_HIP_LIBRARY: a variant build of the same library (kernel experiments, tools/variant_lib.sh); still a HIP library - no fallback
    path = os.environ.get("SLSLAM_HIP_LIBRARY") or LIB_PATH
    if not os.path.exists(path):
        raise SlslamError(2, "libslslam_hip.so not built (run __graft_entry__.build()): " + path)
    L = C.CDLL(path)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.slslam_default_options.argtypes = [C.POINTER(SolverOptions)]
    L.slslam_default_options.restype = None
    L.slslam_lba_solve.argtypes = [C.POINTER(LBAWindow), C.POINTER(SolverOptions), C.POINTER(Summary),
                                   C.POINTER(Iteration), C.c_int, ip]
    L.slslam_lba_batch_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.slslam_lba_batch_destroy.argtypes = [vp]
    L.slslam_lba_batch_destroy.restype = None
    L.slslam_lba_batch_add.argtypes = [vp, C.POINTER(LBAWindow), ip]
    L.slslam_lba_batch_finalize.argtypes = [vp, C.POINTER(SolverOptions)]
    L.slslam_lba_batch_solve.argtypes = [vp, vp]
    L.slslam_lba_batch_reset.argtypes = [vp, vp]
    L.slslam_lba_batch_download.argtypes = [vp, vp]
    L.slslam_lba_batch_download_async.argtypes = [vp, vp]
    L.slslam_lba_batch_wait.argtypes = [vp]
    L.slslam_lba_batch_refill.argtypes = [vp, C.POINTER(LBAWindow), C.c_int, vp]
    L.slslam_lba_stream_create.argtypes = [C.c_int, C.POINTER(SolverOptions), C.c_int, C.POINTER(vp)]
    L.slslam_lba_stream_destroy.argtypes = [vp]
    L.slslam_lba_stream_destroy.restype = None
    L.slslam_lba_stream_submit.argtypes = [vp, C.POINTER(LBAWindow), C.c_int, ip]
    L.slslam_lba_stream_collect.argtypes = [vp, C.c_int, C.POINTER(Summary)]
    L.slslam_lba_stream_stats.argtypes = [vp, dp, dp, dp] + [C.POINTER(C.c_longlong)] * 4 + [ip]
    L.slslam_lba_stream_build_stats.argtypes = [vp] + [C.POINTER(C.c_longlong)] * 3
    L.slslam_lba_stream_submit_packed.argtypes = [vp, C.POINTER(LBAWindow), C.POINTER(C.POINTER(C.c_uint)), C.c_int, ip]
    L.slslam_lba_stream_batch.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.slslam_pinned_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.slslam_pinned_free.argtypes = [vp]
    L.slslam_pinned_register.argtypes = [vp, C.c_size_t]
    L.slslam_pinned_unregister.argtypes = [vp]
    L.slslam_pinned_contains.argtypes = [vp, C.c_size_t]
    L.slslam_pack_indices.argtypes = [C.c_int, ip, ip, ip, C.POINTER(C.c_uint)]
    L.slslam_debug_device_pack.argtypes = [C.POINTER(LBAWindow), C.c_int, ip, ip, ip, ip, ip, ip, C.POINTER(C.c_ubyte), ip, C.c_int, C.c_int,
                                           C.POINTER(C.c_ushort), C.POINTER(C.c_uint), ip]
    L.slslam_debug_device_pack_timed.argtypes = L.slslam_debug_device_pack.argtypes + [C.POINTER(C.c_ulonglong)]
    L.slslam_lba_batch_get_parameters.argtypes = [vp, C.c_int, dp]
    L.slslam_lba_batch_get_summary.argtypes = [vp, C.c_int, C.POINTER(Summary)]
    L.slslam_lba_batch_get_trace.argtypes = [vp, C.c_int, C.POINTER(Iteration), C.c_int, ip]
    L.slslam_lba_batch_export_device.argtypes = [vp, vp, vp]
    L.slslam_lba_batch_counts.argtypes = [vp] + [C.POINTER(C.c_longlong)] * 5
    L.slslam_lba_batch_window_chunks.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.slslam_lba_batch_path.argtypes = [vp, C.POINTER(C.c_int)]
    L.slslam_lba_batch_elimination.argtypes = [vp, C.POINTER(C.c_int)]
    L.slslam_lba_batch_iterations.argtypes = [vp, vp, C.POINTER(C.c_longlong), C.c_int]
    L.slslam_lba_batch_set_profiling.argtypes = [vp, C.c_int]
    L.slslam_lba_batch_kernel_times.argtypes = [vp, dp, ip]
    L.slslam_lba_batch_linearise.argtypes = [vp, C.c_int, dp, dp, dp, dp]
    L.slslam_po_solve.argtypes = [C.POINTER(POGraph), C.POINTER(SolverOptions), C.POINTER(Summary),
                                  C.POINTER(Iteration), C.c_int, ip]
    L.slslam_ransac_score.argtypes = [C.POINTER(RansacFrame), C.c_double, C.c_double, ip, C.POINTER(C.c_ulonglong)]
    L.slslam_po_structure.argtypes = [C.POINTER(POGraph), ip, C.c_int, ip, ip, ip, ip, ip, ip, ip]
    L.slslam_ransac_generate.argtypes = [C.POINTER(RansacTrials), C.c_double, dp, ip]
    L.slslam_ransac_motion.argtypes = [C.POINTER(RansacTrials), dp, C.c_double, C.c_double, C.c_double, C.c_int, ip, ip, dp,
                                       C.POINTER(C.c_ulonglong)]
    L.slslam_ransac_motion_batch.argtypes = [C.c_int, C.POINTER(RansacTrials), C.POINTER(dp), C.c_double, C.c_double, C.c_double,
                                             C.c_int, ip, ip, dp, C.POINTER(C.POINTER(C.c_ulonglong))]
    L.slslam_po_set_profiling.argtypes = [C.c_int]
    L.slslam_po_last_timing.argtypes = [dp, dp, ip, ip, ip]
    L.slslam_debug_phase_cycles.argtypes = [vp, dp]
    L.slslam_debug_read_cycles.argtypes = [vp, C.POINTER(C.c_ulonglong), C.c_longlong, C.POINTER(C.c_longlong)]
    L.slslam_device_count.restype = C.c_int
    L.slslam_release_cached_memory.restype = None
    L.slslam_version.restype = C.c_char_p
    L.slslam_status_string.argtypes = [C.c_int]
    L.slslam_status_string.restype = C.c_char_p
    _lib = L
    return L


def _check(status, where):
    if status != OK:
        raise SlslamError(status, where)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_options(**kw):
    o = SolverOptions()
    lib().slslam_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown solver option %r" % k)
        setattr(o, k, v)
    return o


def release_cached_memory():
    """Frees the device block the one-shot solves of this thread keep between calls."""
    lib().slslam_release_cached_memory()


def device_count():
    return lib().slslam_device_count()


def _summary_dict(s):
    d = {k: getattr(s, k) for k, _ in Summary._fields_}
    d["termination"] = TERMINATION.get(d["termination_type"], "?")
    return d


def _trace_list(tr, n):
    return [{k: getattr(tr[i], k) for k, _ in Iteration._fields_} for i in range(n)]


class PinnedArena:
    """One block of page-locked host memory (slslam_pinned_alloc) that numpy arrays are carved out of: what a caller that streams windows
    allocates its five arrays per window from (instead of `new[]`, reference src/slam.cpp:899-903), so that the GPU reads them in place."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes) + 64
        _check(lib().slslam_pinned_alloc(self.nbytes, C.byref(self.ptr)), "slslam_pinned_alloc")
        self._buf = (C.c_ubyte * self.nbytes).from_address(self.ptr.value)
        self._mem = np.frombuffer(self._buf, dtype=np.uint8)
        self._off = (-self.ptr.value) % 64

    def take(self, a):
        """A copy of array `a` inside the block (64-byte aligned)."""
        a = np.ascontiguousarray(a)
        n = a.nbytes
        if self._off + n > self.nbytes:
            raise MemoryError("PinnedArena exhausted")
        out = self._mem[self._off:self._off + n].view(a.dtype).reshape(a.shape)
        out[...] = a
        self._off += (n + 63) & ~63
        return out

    def close(self):
        if self.ptr:
            self._mem = None
            self._buf = None
            lib().slslam_pinned_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _WindowArrays:
    """Keeps the numpy buffers a slslam_lba_window points to alive."""

    def __init__(self, w, params=None, arena=None, params_arena=None, obs_arena=None):
        self.cam = np.ascontiguousarray(w["camera_index"], dtype=np.int32)
        self.line = np.ascontiguousarray(w["line_index"], dtype=np.int32)
        self.fixed = np.ascontiguousarray(w["fixed_index"], dtype=np.int32).reshape(-1)
        self.obs = np.ascontiguousarray(w["observations"], dtype=np.float64).reshape(-1)
        self.params = np.array(w["parameters"] if params is None else params, dtype=np.float64).reshape(-1).copy()
        if arena is not None:
            self.cam, self.line, self.fixed = (arena.take(x) for x in (self.cam, self.line, self.fixed))
            self.obs = (obs_arena or arena).take(self.obs)
        if params_arena is not None or arena is not None:
            self.params = (params_arena or arena).take(self.params)
        m = len(self.cam)
        if len(self.line) != m or len(self.fixed) != 2 * m or len(self.obs) != 8 * m:
            raise ValueError("inconsistent window arrays")
        if len(self.params) != 6 * int(w["num_cameras"]) + 4 * int(w["num_lines"]):
            raise ValueError("parameter vector has the wrong length")
        self.c = LBAWindow(int(w["num_cameras"]), int(w["num_lines"]), m, _ip(self.cam), _ip(self.line),
                           _ip(self.fixed), _dp(self.obs), _dp(self.params))


def lba_solve(w, params=None, trace_cap=64, **opt):
    """One window through slslam_lba_solve (LBAProblem::build + set_options + ceres::Solve).
    Returns (solved parameters, summary dict, trace list)."""
    arr = _WindowArrays(w, params)
    o = default_options(**opt)
    s = Summary()
    tr = (Iteration * trace_cap)()
    n = C.c_int(0)
    _check(lib().slslam_lba_solve(C.byref(arr.c), C.byref(o), C.byref(s), tr, trace_cap, C.byref(n)), "slslam_lba_solve")
    return arr.params, _summary_dict(s), _trace_list(tr, min(n.value, trace_cap))


class LBABatch:
    """Many independent windows resident in HBM, solved in lock-step (slslam_lba_batch_*)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        _check(lib().slslam_lba_batch_create(int(device), C.byref(self._h)), "slslam_lba_batch_create")
        self.sizes = []          # (num_cameras, num_lines) per window
        self.finalized = False

    def close(self):
        if self._h:
            lib().slslam_lba_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, w, params=None):
        arr = _WindowArrays(w, params)
        idx = C.c_int(-1)
        _check(lib().slslam_lba_batch_add(self._h, C.byref(arr.c), C.byref(idx)), "slslam_lba_batch_add")
        self.sizes.append((int(w["num_cameras"]), int(w["num_lines"])))
        return idx.value

    def finalize(self, **opt):
        o = default_options(**opt)
        _check(lib().slslam_lba_batch_finalize(self._h, C.byref(o)), "slslam_lba_batch_finalize")
        self.finalized = True
        self.options = o

    def solve(self, stream=None):
        _check(lib().slslam_lba_batch_solve(self._h, C.c_void_p(stream or 0)), "slslam_lba_batch_solve")

    def reset(self, stream=None):
        _check(lib().slslam_lba_batch_reset(self._h, C.c_void_p(stream or 0)), "slslam_lba_batch_reset")

    def download(self, stream=None):
        _check(lib().slslam_lba_batch_download(self._h, C.c_void_p(stream or 0)), "slslam_lba_batch_download")

    def download_async(self, stream=None):
        _check(lib().slslam_lba_batch_download_async(self._h, C.c_void_p(stream or 0)), "slslam_lba_batch_download_async")

    def wait(self):
        _check(lib().slslam_lba_batch_wait(self._h), "slslam_lba_batch_wait")

    def refill(self, windows, stream=None):
        """Replaces every window of the finalized batch (slslam_lba_batch_refill); `windows`: a WindowSet or a list of window dicts."""
        ws = windows if isinstance(windows, WindowSet) else WindowSet(windows)
        _check(lib().slslam_lba_batch_refill(self._h, ws.c, len(ws), C.c_void_p(stream or 0)), "slslam_lba_batch_refill")
        self.sizes = list(ws.sizes)

    def export_device(self, device_ptr, stream=None):
        _check(lib().slslam_lba_batch_export_device(self._h, C.c_void_p(device_ptr), C.c_void_p(stream or 0)),
               "slslam_lba_batch_export_device")

    def parameters(self, i):
        c, l = self.sizes[i]
        out = np.empty(6 * c + 4 * l)
        _check(lib().slslam_lba_batch_get_parameters(self._h, i, _dp(out)), "slslam_lba_batch_get_parameters")
        return out

    def summary(self, i):
        s = Summary()
        _check(lib().slslam_lba_batch_get_summary(self._h, i, C.byref(s)), "slslam_lba_batch_get_summary")
        return _summary_dict(s)

    def trace(self, i, cap=64):
        tr = (Iteration * cap)()
        n = C.c_int(0)
        _check(lib().slslam_lba_batch_get_trace(self._h, i, tr, cap, C.byref(n)), "slslam_lba_batch_get_trace")
        return _trace_list(tr, min(n.value, cap))

    def counts(self):
        v = [C.c_longlong(0) for _ in range(5)]
        _check(lib().slslam_lba_batch_counts(self._h, *[C.byref(x) for x in v]), "slslam_lba_batch_counts")
        return dict(zip(["windows", "cameras", "free_cameras", "lines", "observations"], [x.value for x in v]))

    def path(self):
        v = C.c_int(-1)
        _check(lib().slslam_lba_batch_path(self._h, C.byref(v)), "slslam_lba_batch_path")
        return v.value

    def elimination(self):
        """The lba_elimination value that reproduces the sweep this batch runs (what the automatic choice resolved to)."""
        v = C.c_int(-1)
        _check(lib().slslam_lba_batch_elimination(self._h, C.byref(v)), "slslam_lba_batch_elimination")
        return v.value

    def window_chunks(self, i):
        v = C.c_int(0)
        _check(lib().slslam_lba_batch_window_chunks(self._h, int(i), C.byref(v)), "slslam_lba_batch_window_chunks")
        return v.value

    def iterations(self, stream=None, clear=False):
        v = C.c_longlong(0)
        _check(lib().slslam_lba_batch_iterations(self._h, C.c_void_p(stream or 0), C.byref(v), int(clear)),
               "slslam_lba_batch_iterations")
        return v.value

    def total_parameters(self):
        return sum(6 * c + 4 * l for c, l in self.sizes)

    def set_profiling(self, enable):
        _check(lib().slslam_lba_batch_set_profiling(self._h, int(bool(enable))), "slslam_lba_batch_set_profiling")

    def kernel_times(self):
        ms = np.zeros(8)
        n = np.zeros(8, dtype=np.int32)
        _check(lib().slslam_lba_batch_kernel_times(self._h, _dp(ms), _ip(n)), "slslam_lba_batch_kernel_times")
        return {KERNEL_FAMILIES[i]: (float(ms[i]), int(n[i])) for i in range(8)}

    def linearise(self, i, num_observations):
        m = int(num_observations)
        r, jc, jl, c = np.zeros((m, 4)), np.zeros((m, 4, 6)), np.zeros((m, 4, 4)), np.zeros(1)
        _check(lib().slslam_lba_batch_linearise(self._h, i, _dp(r), _dp(jc), _dp(jl), _dp(c)), "slslam_lba_batch_linearise")
        return float(c[0]), r, jc, jl


class WindowSet:
    """A C array of slslam_lba_window over numpy buffers that stay alive with it (what a caller of the stream / refill entry points
    holds: the five arrays of every window, reference src/slam.cpp:899-921)."""

    def __init__(self, windows, pinned=False, packed=False):
        """pinned: the arrays live in page-locked blocks (slslam_pinned_alloc): the device build reads them in place and the solved
        parameters are written back into them by the GPU.  packed: the three index arrays of every window are ALSO held narrowed to one
        32-bit word per observation (slslam_pack_indices), and LBAStream.submit hands those over instead (slslam_lba_stream_submit_packed)."""
        self.arena = self.params_arena = self.obs_arena = None
        self.packed = None
        if pinned:
            # three blocks - the index arrays of all windows, their observation arrays, their parameter arrays (derive() gives a set parameter
            # arrays of its own over the same inputs): arrays that lie next to each other go up in a few large copies of the copy engine, and the
            # index arrays - which the host threads narrow on the way - do not sit between the observation arrays
            self.arena = PinnedArena(sum(16 * len(w["camera_index"]) + 3 * 64 for w in windows) + 4096)
            self.obs_arena = PinnedArena(sum(64 * len(w["camera_index"]) + 64 for w in windows) + 4096)
            self.params_arena = PinnedArena(sum(8 * (6 * int(w["num_cameras"]) + 4 * int(w["num_lines"])) + 64 for w in windows) + 4096)
        if pinned and packed:
            self.arena.close()
            self.arena = PinnedArena(sum(4 * len(w["camera_index"]) + 64 for w in windows) + 4096)
        self.arrays = [_WindowArrays(w, arena=None if packed else self.arena, params_arena=self.params_arena, obs_arena=self.obs_arena) for w in windows]
        if packed:
            self.packed_arrays = []
            for a in self.arrays:
                pk = np.zeros(len(a.cam), dtype=np.uint32)
                _check(lib().slslam_pack_indices(len(a.cam), _ip(a.cam), _ip(a.line), _ip(a.fixed), pk.ctypes.data_as(C.POINTER(C.c_uint))), "slslam_pack_indices")
                if self.arena is not None:
                    pk = self.arena.take(pk)
                    a.obs = self.obs_arena.take(a.obs)
                    a.c = LBAWindow(a.c.num_cameras, a.c.num_lines, a.c.num_observations, _ip(a.cam), _ip(a.line), _ip(a.fixed), _dp(a.obs), _dp(a.params))
                self.packed_arrays.append(pk)
            self.packed = (C.POINTER(C.c_uint) * max(len(self.arrays), 1))(*[p_.ctypes.data_as(C.POINTER(C.c_uint)) for p_ in self.packed_arrays])
        self.c = (LBAWindow * max(len(self.arrays), 1))(*[a.c for a in self.arrays])
        self.sizes = [(int(w["num_cameras"]), int(w["num_lines"])) for w in windows]

    def __len__(self):
        return len(self.arrays)

    def parameters(self, i):
        return self.arrays[i].params

    def derive(self, order):
        """Another set over the SAME input arrays (indices, observations: read only) in another window order, with parameter arrays of its
        own holding the present values of this set's - what a bench needs to submit many distinct sets without holding each set's gigabyte
        of observations once more.  Page-locked like this set."""
        import copy
        out = WindowSet.__new__(WindowSet)
        out.arena = out.params_arena = out.obs_arena = None
        out.packed = None
        if self.packed is not None:
            out.packed_arrays = [self.packed_arrays[j] for j in order]
            out.packed = (C.POINTER(C.c_uint) * max(len(order), 1))(*[p_.ctypes.data_as(C.POINTER(C.c_uint)) for p_ in out.packed_arrays])
        if self.arena is not None:
            out.params_arena = PinnedArena(sum(a.params.nbytes + 64 for a in self.arrays) + 4096)
        out.arrays = []
        for j in order:
            a = copy.copy(self.arrays[j])
            a.params = out.params_arena.take(a.params) if out.params_arena is not None else a.params.copy()
            a.c = LBAWindow(a.c.num_cameras, a.c.num_lines, a.c.num_observations, _ip(a.cam), _ip(a.line), _ip(a.fixed), _dp(a.obs), _dp(a.params))
            out.arrays.append(a)
        out.c = (LBAWindow * max(len(out.arrays), 1))(*[a.c for a in out.arrays])
        out.sizes = [self.sizes[j] for j in order]
        out._base = self                      # (the shared arrays live in the base set's block)
        return out

    def close(self):
        self.arrays = [] if (self.arena is not None or self.params_arena is not None) else self.arrays
        for name in ("arena", "params_arena", "obs_arena"):
            if getattr(self, name, None) is not None:
                getattr(self, name).close()
                setattr(self, name, None)


class _BatchView(LBABatch):
    """A batch owned by somebody else (a stream's slot): the getters of LBABatch, no destroy."""

    def __init__(self, handle, sizes):
        self._h = handle
        self.sizes = list(sizes)
        self.finalized = True

    def close(self):
        self._h = C.c_void_p()


def debug_device_pack(w, grouping=0, clocks=None):
    """slslam_debug_device_pack: one window through the device build alone; returns (status bits, dict as tests/test_host_side.py::_pack).
    clocks: a numpy uint64[16] that receives the shader-clock stamps of k_build_window's phases."""
    arr = _WindowArrays(w)
    Cn, L, M = int(w["num_cameras"]), int(w["num_lines"]), len(arr.cam)
    counts = np.zeros(5, dtype=np.int32)
    lo, lp = np.zeros(max(L, 1), dtype=np.int32), np.zeros(L + 1, dtype=np.int32)
    oo, oc, cf = np.zeros(max(M, 1), dtype=np.int32), np.zeros(max(M, 1), dtype=np.int32), np.zeros(max(Cn, 1), dtype=np.int32)
    max_tiles, max_items = L + 8, 64 * M + 8
    tiles = np.zeros(4 * max_tiles, dtype=np.int32)
    items = np.zeros(2 * max_items, dtype=np.uint8)
    lane_map = np.zeros(64 * max_tiles, dtype=np.uint16)
    desc = np.zeros(max(L, 1), dtype=np.uint32)
    st = C.c_int(-1)
    _check(lib().slslam_debug_device_pack_timed(C.byref(arr.c), int(grouping), _ip(counts), _ip(lo), _ip(lp), _ip(oo), _ip(oc), _ip(tiles),
                                                items.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(cf), max_tiles, max_items,
                                                lane_map.ctypes.data_as(C.POINTER(C.c_ushort)), desc.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(st),
                                                clocks.ctypes.data_as(C.POINTER(C.c_ulonglong)) if clocks is not None else None),
           "slslam_debug_device_pack")
    return st.value, dict(desc=desc[:L], Cf=int(counts[0]), ntiles=int(counts[1]), nitems=int(counts[2]), nfree=int(counts[3]), nkept=int(counts[4]),
                          line_order=lo[:L], line_ptr=lp, ob_orig=oo[:M], ob_cam=oc[:M], cam_cf=cf[:Cn],
                          tiles=tiles[:4 * int(counts[1])].reshape(-1, 4), items=items[:2 * int(counts[2])].reshape(-1, 2),
                          lane_map=lane_map[:64 * int(counts[1])].reshape(-1, 64))


class LBAStream:
    """slslam_lba_stream_*: `depth` refillable batches in flight; submit(WindowSet) -> ticket, collect(ticket) -> summaries, the solved
    parameters land in the WindowSet's parameter arrays."""

    def __init__(self, device=-1, depth=3, **opt):
        self._h = C.c_void_p()
        o = default_options(**opt)
        _check(lib().slslam_lba_stream_create(int(device), C.byref(o), int(depth), C.byref(self._h)), "slslam_lba_stream_create")
        self._live = {}

    def close(self):
        if self._h:
            lib().slslam_lba_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, ws):
        t = C.c_int(-1)
        if getattr(ws, "packed", None) is not None:
            _check(lib().slslam_lba_stream_submit_packed(self._h, ws.c, ws.packed, len(ws), C.byref(t)), "slslam_lba_stream_submit_packed")
        else:
            _check(lib().slslam_lba_stream_submit(self._h, ws.c, len(ws), C.byref(t)), "slslam_lba_stream_submit")
        self._live[t.value] = ws
        return t.value

    def collect(self, ticket, want_summaries=True):
        ws = self._live.pop(ticket, None)
        if ws is None:                                   # not a ticket in flight: the library says so
            _check(lib().slslam_lba_stream_collect(self._h, int(ticket), None), "slslam_lba_stream_collect")
            raise SlslamError(5, "slslam_lba_stream_collect")
        sm = (Summary * max(len(ws), 1))() if want_summaries else None
        _check(lib().slslam_lba_stream_collect(self._h, int(ticket), sm), "slslam_lba_stream_collect")
        return [_summary_dict(sm[i]) for i in range(len(ws))] if want_summaries else None

    def batch_of(self, ticket, ws):
        """The batch that served `ticket` (its traces, chunk cuts, sweep): valid until the slot is submitted to again."""
        h = C.c_void_p()
        _check(lib().slslam_lba_stream_batch(self._h, int(ticket), C.byref(h)), "slslam_lba_stream_batch")
        return _BatchView(h, ws.sizes)

    def build_stats(self):
        q = [C.c_longlong(0) for _ in range(3)]
        _check(lib().slslam_lba_stream_build_stats(self._h, *[C.byref(x) for x in q]), "slslam_lba_stream_build_stats")
        return {"device_builds": q[0].value, "zero_copy": q[1].value, "fallback_windows": q[2].value}

    def stats(self):
        d = [C.c_double(0) for _ in range(3)]
        q = [C.c_longlong(0) for _ in range(4)]
        t = C.c_int(0)
        _check(lib().slslam_lba_stream_stats(self._h, *[C.byref(x) for x in d], *[C.byref(x) for x in q], C.byref(t)), "slslam_lba_stream_stats")
        return {"ms_submit": d[0].value, "ms_collect_wait": d[1].value, "ms_collect_copy": d[2].value,
                "refills": q[0].value, "builds": q[1].value, "windows": q[2].value, "lm_iterations": q[3].value, "host_threads": t.value}


def po_solve(g, params=None, trace_cap=64, **opt):
    """One pose graph through slslam_po_solve (POProblem::build + set_options + ceres::Solve)."""
    i1 = np.ascontiguousarray(g["pose_index_1"], dtype=np.int32)
    i2 = np.ascontiguousarray(g["pose_index_2"], dtype=np.int32)
    cons = np.ascontiguousarray(g["constraints"], dtype=np.float64).reshape(-1)
    x = np.array(g["parameters"] if params is None else params, dtype=np.float64).reshape(-1).copy()
    if len(i2) != len(i1) or len(cons) != 6 * len(i1) or len(x) != 6 * int(g["num_poses"]):
        raise ValueError("inconsistent pose-graph arrays")
    cg = POGraph(int(g["num_poses"]), len(i1), _ip(i1), _ip(i2), _dp(cons), _dp(x))
    o = default_options(**opt)
    s = Summary()
    tr = (Iteration * trace_cap)()
    n = C.c_int(0)
    _check(lib().slslam_po_solve(C.byref(cg), C.byref(o), C.byref(s), tr, trace_cap, C.byref(n)), "slslam_po_solve")
    return x, _summary_dict(s), _trace_list(tr, min(n.value, trace_cap))


def po_solve_timed(g, **opt):
    """po_solve with the device-side split: returns (x, summary, dict(total_ms, factor_ms = the slowest factorisation,
    factor_calls, unknowns, junction_unknowns))."""
    lib().slslam_po_set_profiling(1)
    try:
        x, s, _ = po_solve(g, **opt)
        tot, fac = C.c_double(0), C.c_double(0)
        nc, nu, nj = C.c_int(0), C.c_int(0), C.c_int(0)
        lib().slslam_po_last_timing(C.byref(tot), C.byref(fac), C.byref(nc), C.byref(nu), C.byref(nj))
    finally:
        lib().slslam_po_set_profiling(0)
    return x, s, dict(total_ms=tot.value, factor_ms=fac.value, factor_calls=nc.value, unknowns=nu.value, junction_unknowns=nj.value)


def ransac_score(poses, observations, lines, baseline=0.12, error_thr=5.0 / 406.05):
    """Scores motion hypotheses against the common lines (slslam_ransac_score).
    poses [H,12] (R row-major, t), observations [K,8], lines [K,6] -> (scores [H], inlier mask [H,K] bool)."""
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 12)
    obs = np.ascontiguousarray(observations, dtype=np.float64).reshape(-1, 8)
    ln = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 6)
    if len(ln) != len(obs):
        raise ValueError("observations and lines must have the same length")
    h, k = len(poses), len(obs)
    words = (k + 63) // 64
    scores = np.zeros(max(h, 1), dtype=np.int32)
    bits = np.zeros(max(h * words, 1), dtype=np.uint64)
    fr = RansacFrame(h, k, _dp(poses), _dp(obs), _dp(ln))
    _check(lib().slslam_ransac_score(C.byref(fr), float(baseline), float(error_thr), _ip(scores),
                                     bits.ctypes.data_as(C.POINTER(C.c_ulonglong))), "slslam_ransac_score")
    mask = np.zeros((h, k), dtype=bool)
    if h and k:
        b = bits[:h * words].reshape(h, words)
        for w in range(words):
            n = min(64, k - 64 * w)
            mask[:, 64 * w:64 * w + n] = ((b[:, w:w + 1] >> np.arange(n, dtype=np.uint64)) & np.uint64(1)).astype(bool)
    return scores[:h], mask


def _trials(obs0, obs1, samples):
    o0 = np.ascontiguousarray(obs0, dtype=np.float64).reshape(-1, 8)
    o1 = np.ascontiguousarray(obs1, dtype=np.float64).reshape(-1, 8)
    smp = np.ascontiguousarray(samples, dtype=np.int32)
    if smp.ndim != 2 or len(o0) != len(o1):
        raise ValueError("samples must be [trials, s]; obs0 and obs1 must have the same length")
    return RansacTrials(smp.shape[0], smp.shape[1], len(o0), _ip(smp), _dp(o0), _dp(o1)), (o0, o1, smp)


def ransac_generate(obs0, obs1, samples, baseline=-0.12):
    """SLAM::vo_angle_axis_approx for every pre-drawn trial (slslam_ransac_generate) -> (poses [H,12], valid [H])."""
    tr, keep = _trials(obs0, obs1, samples)
    h = tr.num_trials
    poses = np.zeros((max(h, 1), 12))
    valid = np.zeros(max(h, 1), dtype=np.int32)
    _check(lib().slslam_ransac_generate(C.byref(tr), float(baseline), _dp(poses), _ip(valid)), "slslam_ransac_generate")
    return poses[:h], valid[:h]


def ransac_motion(obs0, obs1, lines, samples, baseline=0.12, error_thr=5.0 / 406.05, prob_free_outliers=0.999,
                  max_trials=1000, best_score=0):
    """SLAM::ransac_motion over a pre-drawn sample sequence (slslam_ransac_motion)
    -> (trial_cnt, best_score, best_pose [12], inlier mask [K])."""
    tr, keep = _trials(obs0, obs1, samples)
    ln = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 6)
    k = tr.num_lines
    if len(ln) != k:
        raise ValueError("lines and observations must have the same length")
    words = (k + 63) // 64
    bs = np.array([best_score], dtype=np.int32)
    tc = np.zeros(1, dtype=np.int32)
    pose = np.zeros(12)
    bits = np.zeros(max(words, 1), dtype=np.uint64)
    _check(lib().slslam_ransac_motion(C.byref(tr), _dp(ln), float(baseline), float(error_thr), float(prob_free_outliers),
                                      int(max_trials), _ip(bs), _ip(tc), _dp(pose), bits.ctypes.data_as(C.POINTER(C.c_ulonglong))),
           "slslam_ransac_motion")
    mask = np.zeros(k, dtype=bool)
    for w in range(words):
        n = min(64, k - 64 * w)
        mask[64 * w:64 * w + n] = ((bits[w] >> np.arange(n, dtype=np.uint64)) & np.uint64(1)).astype(bool)
    return int(tc[0]), int(bs[0]), pose, mask


def po_structure(g, max_chains=4096):
    """Symbolic analysis of the structured pose-graph factorisation (slslam_po_structure; host only).
    Returns dict(slot [N], chains [(start, len, left, right)], num_chain_unknowns, num_unknowns)."""
    i1 = np.ascontiguousarray(g["pose_index_1"], dtype=np.int32)
    i2 = np.ascontiguousarray(g["pose_index_2"], dtype=np.int32)
    n = int(g["num_poses"])
    cg = POGraph(n, len(i1), _ip(i1), _ip(i2), None, None)
    slot = np.zeros(max(n, 1), dtype=np.int32)
    arr = [np.zeros(max_chains, dtype=np.int32) for _ in range(4)]
    nc, ncu, nu = (np.zeros(1, dtype=np.int32) for _ in range(3))
    _check(lib().slslam_po_structure(C.byref(cg), _ip(slot), max_chains, _ip(nc), _ip(arr[0]), _ip(arr[1]), _ip(arr[2]), _ip(arr[3]),
                                     _ip(ncu), _ip(nu)), "slslam_po_structure")
    k = int(nc[0])
    return {"slot": slot[:n], "chains": [tuple(int(a[c]) for a in arr) for c in range(k)],
            "num_chain_unknowns": int(ncu[0]), "num_unknowns": int(nu[0]), "level1_chains": int(lib().slslam_po_structure_level1())}


def ransac_motion_batch(frames, baseline=0.12, error_thr=5.0 / 406.05, prob_free_outliers=0.999, max_trials=1000):
    """slslam_ransac_motion_batch: frames = list of dicts with obs0, obs1, lines, samples (as make_ransac_pair returns).
    Returns a list of (trial_cnt, best_score, best_pose [12], inlier mask [K]) per frame."""
    n = len(frames)
    keep, trs, lns, bitbufs = [], (RansacTrials * max(n, 1))(), (C.POINTER(C.c_double) * max(n, 1))(), []
    bitptrs = (C.POINTER(C.c_ulonglong) * max(n, 1))()
    for i, fr in enumerate(frames):
        tr, k = _trials(fr["obs0"], fr["obs1"], fr["samples"])
        ln = np.ascontiguousarray(fr["lines"], dtype=np.float64).reshape(-1, 6)
        keep.append((k, ln))
        trs[i] = tr
        lns[i] = _dp(ln)
        bits = np.zeros(max((tr.num_lines + 63) // 64, 1), dtype=np.uint64)
        bitbufs.append(bits)
        bitptrs[i] = bits.ctypes.data_as(C.POINTER(C.c_ulonglong))
    bs = np.zeros(max(n, 1), dtype=np.int32)
    tc = np.zeros(max(n, 1), dtype=np.int32)
    poses = np.zeros((max(n, 1), 12))
    _check(lib().slslam_ransac_motion_batch(n, trs, lns, float(baseline), float(error_thr), float(prob_free_outliers), int(max_trials),
                                            _ip(bs), _ip(tc), _dp(poses), bitptrs), "slslam_ransac_motion_batch")
    out = []
    for i in range(n):
        k = trs[i].num_lines
        mask = np.zeros(k, dtype=bool)
        for w in range((k + 63) // 64):
            m = min(64, k - 64 * w)
            mask[64 * w:64 * w + m] = ((bitbufs[i][w] >> np.arange(m, dtype=np.uint64)) & np.uint64(1)).astype(bool)
        out.append((int(tc[i]), int(bs[i]), poses[i].copy(), mask))
    return out
