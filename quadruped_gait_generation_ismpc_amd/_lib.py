"""ctypes binding of the C ABI (include/ismpc.h).  Fails loudly when the HIP library is missing:
there is no Python or CPU fallback for the hot path."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ISMPC_LIB") or os.path.join(PKG, "libismpc_hip.so")     # ISMPC_LIB: a tuning sweep's variant build


class Params(C.Structure):
    """ismpc_params: parameters.cpp:9-45 and MPCSolver.cpp:253-255 as run-time values."""
    _fields_ = [("N", C.c_int32), ("S", C.c_int32), ("F", C.c_int32), ("M", C.c_int32),
                ("mpc_dt", C.c_double), ("control_dt", C.c_double), ("mass", C.c_double),
                ("g", C.c_double), ("h_des", C.c_double), ("foot_width", C.c_double),
                ("first_step_halfwidth", C.c_double),
                ("q_p", C.c_double), ("q_u", C.c_double), ("q_v", C.c_double),
                ("z_ineq_lo", C.c_double), ("z_ineq_hi", C.c_double),
                ("lambda_gate", C.c_double)]


TICK_IN = np.dtype([("com_pos", "<f8", 3), ("com_vel", "<f8", 3), ("simulation_time", "<f8"),
                    ("mpc_iter", "<i4"), ("control_iter", "<i4"), ("footstep_counter", "<i4"),
                    ("reserved", "<i4")], align=False)
TICK_OUT = np.dtype([("com_pos", "<f8", 3), ("com_vel", "<f8", 3), ("u0", "<f8", 3),
                     ("status", "<i4"), ("iters", "<i4")], align=False)
assert TICK_IN.itemsize == 72 and TICK_OUT.itemsize == 80

# every symbol include/ismpc.h declares
EXPORTS = ["ismpc_params_default", "ismpc_create", "ismpc_destroy", "ismpc_solve_batch",
           "ismpc_solve_batch_device", "ismpc_rollout_device", "ismpc_abi_version", "ismpc_last_error",
           "ismpc_get_params", "ismpc_midpoint_rows", "ismpc_get_midpoint", "ismpc_set_timing",
           "ismpc_last_kernel_ms", "ismpc_reserve", "ismpc_host_alloc", "ismpc_host_free", "ismpc_host_register",
           "ismpc_host_unregister", "ismpc_create_sweep", "ismpc_sweep_info", "ismpc_sweep_verify_tables", "ismpc_fallback_counters", "ismpc_sweep_bind"]

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} is not built (run `python -c 'import __graft_entry__ as g; g.build()'`): "
            "the ISMPC hot path is HIP-only and has no CPU fallback")
    # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).
    # Whichever is mapped first serves both, so torch -- which owns device memory and streams on the
    # Python side -- must be imported BEFORE this library; loaded the other way round the process
    # ends up with two HSA runtimes and neither sees the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    lib.ismpc_params_default.argtypes = [C.POINTER(Params)]; lib.ismpc_params_default.restype = None
    lib.ismpc_create.argtypes = [C.POINTER(Params), vp, ci, ci, C.POINTER(vp)]; lib.ismpc_create.restype = ci
    lib.ismpc_destroy.argtypes = [vp]; lib.ismpc_destroy.restype = None
    lib.ismpc_solve_batch.argtypes = [vp, ci, vp, vp]; lib.ismpc_solve_batch.restype = ci
    lib.ismpc_solve_batch_device.argtypes = [vp, ci, vp, vp, vp, vp]; lib.ismpc_solve_batch_device.restype = ci
    lib.ismpc_rollout_device.argtypes = [vp, ci, vp, ci, ci, vp, vp]; lib.ismpc_rollout_device.restype = ci
    lib.ismpc_abi_version.argtypes = []; lib.ismpc_abi_version.restype = ci
    lib.ismpc_last_error.argtypes = []; lib.ismpc_last_error.restype = C.c_char_p
    lib.ismpc_get_params.argtypes = [vp, C.POINTER(Params)]; lib.ismpc_get_params.restype = ci
    lib.ismpc_midpoint_rows.argtypes = [vp]; lib.ismpc_midpoint_rows.restype = ci
    lib.ismpc_get_midpoint.argtypes = [vp, vp, ci]; lib.ismpc_get_midpoint.restype = ci
    lib.ismpc_set_timing.argtypes = [vp, ci]; lib.ismpc_set_timing.restype = ci
    lib.ismpc_last_kernel_ms.argtypes = [vp]; lib.ismpc_last_kernel_ms.restype = cd
    lib.ismpc_reserve.argtypes = [vp, ci]; lib.ismpc_reserve.restype = ci
    lib.ismpc_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]; lib.ismpc_host_alloc.restype = ci
    lib.ismpc_host_free.argtypes = [vp]; lib.ismpc_host_free.restype = ci
    lib.ismpc_host_register.argtypes = [vp, C.c_size_t]; lib.ismpc_host_register.restype = ci
    lib.ismpc_host_unregister.argtypes = [vp]; lib.ismpc_host_unregister.restype = ci
    lib.ismpc_create_sweep.argtypes = [vp, ci, vp, ci, ci, C.POINTER(vp)]; lib.ismpc_create_sweep.restype = ci
    lib.ismpc_sweep_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(cd)]; lib.ismpc_sweep_info.restype = ci
    lib.ismpc_sweep_verify_tables.argtypes = [vp, ci, vp]; lib.ismpc_sweep_verify_tables.restype = ci
    lib.ismpc_fallback_counters.argtypes = [vp, vp]; lib.ismpc_fallback_counters.restype = ci
    lib.ismpc_sweep_bind.argtypes = [vp, ci, vp, vp]; lib.ismpc_sweep_bind.restype = ci
    _lib = lib
    return lib


def last_error():
    return load().ismpc_last_error().decode()


class PinnedRecords:
    """A numpy array of records in page-locked host memory (ismpc_host_alloc): what the zero-copy host entry point
    ismpc_solve_batch wants on both sides.  `.array` is the view.  The block is released when the LAST numpy view of it dies
    (the finalizer is keyed to the buffer object every view's `.base` chain ends in), so a slice or a reshape that outlives this
    object, or a call to free(), never points at freed memory."""

    def __init__(self, n, dtype):
        import weakref
        lib = load()
        p = C.c_void_p()
        nbytes = max(int(n), 1) * np.dtype(dtype).itemsize
        rc = lib.ismpc_host_alloc(nbytes, C.byref(p))
        if rc != 0:
            raise MemoryError(f"ismpc_host_alloc({nbytes}): {rc}: {last_error()}")
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        weakref.finalize(buf, lib.ismpc_host_free, C.c_void_p(p.value))
        self.array = np.frombuffer(buf, dtype=dtype, count=int(n))

    def free(self):
        """Drops this object's view; the memory goes when no other view of it is left."""
        self.array = None
