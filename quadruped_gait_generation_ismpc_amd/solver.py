"""Host-side mirror of the reference's MPCSolver interface (AMR_code_DART/MPCSolver.hpp:16-28)
over the C ABI of include/ismpc.h.  Same names and argument meaning as the reference:

    solver = MPCSolver(ftsp_and_timings)                    # MPCSolver.cpp:5
    next   = solver.solve(current, walkState, ftsp_and_timings)   # MPCSolver.cpp:204

plus the batch entry points the data-parallel axis needs.  Every compute call goes to the HIP
library; nothing here computes a tick in Python.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import Params, TICK_IN, TICK_OUT

ST_X_INFEASIBLE, ST_Y_INFEASIBLE, ST_Z_INEQ_ACTIVE, ST_BAD_INDEX = 1, 2, 4, 8
ST_FLIGHT, ST_TICK_SKIPPED, ST_Z_NAN, ST_Z_FAILED = 16, 32, 64, 128
ST_ERROR_MASK = ST_X_INFEASIBLE | ST_Y_INFEASIBLE | ST_BAD_INDEX | ST_Z_FAILED


class IsmpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ismpc error {code}: {msg}")
        self.code = code


def default_params(**overrides):
    """The reference's constants (parameters.cpp:9-45, MPCSolver.cpp:253-255) from ismpc_params_default."""
    p = Params()
    _lib.load().ismpc_params_default(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def reference_plan(rows=40, params=None):
    """ftsp_and_time exactly as Controller.cpp:89-97 builds it (row 0 stays zero)."""
    p = params if params is not None else default_params()
    ftsp = np.zeros((rows, 4))
    for i in range(1, rows):
        ftsp[i, 0] = (i - 1) * 0.2
        ftsp[i, 1] = (-1.0) ** (i - 1) * 0.08
        ftsp[i, 2] = 0.0
        ftsp[i, 3] = (p.mpc_dt / p.control_dt) * (p.S + p.F) * i
    return ftsp


@dataclass
class State:
    """The fields of types.hpp:7-28 that MPCSolver::solve reads or writes; everything else the
    reference's State carries passes through solve() untouched (next = current, MPCSolver.cpp:210)."""
    comPos: np.ndarray = field(default_factory=lambda: np.zeros(3))
    comVel: np.ndarray = field(default_factory=lambda: np.zeros(3))
    comAcc: np.ndarray = field(default_factory=lambda: np.zeros(3))
    zmpPos: np.ndarray = field(default_factory=lambda: np.zeros(3))


@dataclass
class WalkState:
    """types.hpp:77-81."""
    supportFoot: bool = True
    simulationTime: float = 0.0
    mpcIter: int = 0
    controlIter: int = 0
    footstepCounter: int = 0
    indInitial: int = 0


class MPCSolver:
    """Batched drop-in for the reference's MPCSolver.  `ftsp_and_timings` is rows x 4 (x, y, z, t)."""

    def __init__(self, ftsp_and_timings, params=None, device=0):
        self._lib = _lib.load()
        self.params = params if params is not None else default_params()
        self.ftsp = np.ascontiguousarray(ftsp_and_timings, dtype=np.float64)
        if self.ftsp.ndim != 2 or self.ftsp.shape[1] != 4:
            raise ValueError("ftsp_and_timings must be rows x 4")
        h = C.c_void_p()
        rc = self._lib.ismpc_create(C.byref(self.params), self.ftsp.ctypes.data_as(C.c_void_p),
                                    self.ftsp.shape[0], int(device), C.byref(h))
        if rc != 0:
            raise IsmpcError(rc, _lib.last_error())
        self._h = h
        self.device = int(device)
        # public diagnostic fields of MPCSolver.hpp:24-28, set as MPCSolver.cpp:98-102,206-207 do
        self.itr = 0; self.fsCount = 0; self.old_fsCount = 0; self.ct = 0
        self.xz_dot = 0.0; self.yz_dot = 0.0

    @classmethod
    def sweep(cls, ftsp_and_timings, params_list, device=0):
        """Parameter sweep (ismpc_create_sweep): one handle for `len(params_list)` parameter sets that share the horizon and the
        plan; instance i runs with set tick_in["reserved"][i].  Every set's tables are built on the device."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.params_list = list(params_list)
        self.params = self.params_list[0]
        self.ftsp = np.ascontiguousarray(ftsp_and_timings, dtype=np.float64)
        if self.ftsp.ndim != 2 or self.ftsp.shape[1] != 4:
            raise ValueError("ftsp_and_timings must be rows x 4")
        if not self.params_list:
            raise ValueError("a sweep needs at least one parameter set")
        arr = (Params * len(self.params_list))(*self.params_list)
        h = C.c_void_p()
        rc = self._lib.ismpc_create_sweep(C.cast(arr, C.c_void_p), len(self.params_list), self.ftsp.ctypes.data_as(C.c_void_p),
                                          self.ftsp.shape[0], int(device), C.byref(h))
        if rc != 0:
            raise IsmpcError(rc, _lib.last_error())
        self._h = h; self.device = int(device)
        self.itr = 0; self.fsCount = 0; self.old_fsCount = 0; self.ct = 0; self.xz_dot = 0.0; self.yz_dot = 0.0
        return self

    def sweep_info(self):
        n, it, gl, ms = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        self._check(self._lib.ismpc_sweep_info(self._h, C.byref(n), C.byref(it), C.byref(gl), C.byref(ms)))
        return {"n_sets": n.value, "newton_iterations": it.value, "mfma_gemm_launches": gl.value, "build_ms": ms.value}

    def sweep_bind(self, tick_in_u8):
        """ismpc_sweep_bind on a CUDA uint8 tensor [batch, 72]: later calls of the same batch size run sorted by parameter set."""
        import torch
        assert tick_in_u8.is_cuda and tick_in_u8.dtype == torch.uint8 and tick_in_u8.shape[1] == 72 and tick_in_u8.is_contiguous()
        s_ = torch.cuda.current_stream(tick_in_u8.device).cuda_stream
        self._check(self._lib.ismpc_sweep_bind(self._h, int(tick_in_u8.shape[0]), C.c_void_p(tick_in_u8.data_ptr()), C.c_void_p(s_) if s_ else None))

    def sweep_unbind(self):
        self._check(self._lib.ismpc_sweep_bind(self._h, 0, None, None))

    def sweep_verify_tables(self, k):
        """max |device - host long double| / max |host| per table kind of parameter set k (ismpc_sweep_verify_tables)."""
        e = np.zeros(8)
        self._check(self._lib.ismpc_sweep_verify_tables(self._h, int(k), e.ctypes.data_as(C.c_void_p)))
        return dict(zip(("Hinv", "affine", "W", "SW", "HSt", "SHSt", "tail", "layout"), e.tolist()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ismpc_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def N(self):
        return self.params.N

    def _check(self, rc):
        if rc != 0:
            raise IsmpcError(rc, _lib.last_error())

    # ---- the reference's own call shape (batch of one) ----
    def solve(self, current, walkState, ftsp_and_timings=None):
        """State MPCSolver::solve(State current, WalkState walkState, const MatrixXd&) -- MPCSolver.cpp:204.
        The third argument is accepted and ignored, as in the reference (only read at :441, result unused)."""
        self.itr = walkState.mpcIter; self.fsCount = walkState.footstepCounter   # MPCSolver.cpp:206-207
        rec = np.zeros(1, dtype=TICK_IN)
        rec["com_pos"][0] = current.comPos; rec["com_vel"][0] = current.comVel
        rec["simulation_time"] = walkState.simulationTime
        rec["mpc_iter"] = walkState.mpcIter; rec["control_iter"] = walkState.controlIter
        rec["footstep_counter"] = walkState.footstepCounter
        out = self.solve_batch(rec)
        nxt = State(comPos=out["com_pos"][0].copy(), comVel=out["com_vel"][0].copy(),
                    comAcc=np.array(current.comAcc, dtype=float), zmpPos=np.array(current.zmpPos, dtype=float))
        return nxt

    # ---- batch, host buffers ----
    def solve_batch(self, tick_in, out=None):
        """Host records in, host records out (ismpc_solve_batch).  With both arrays in page-locked memory (PinnedRecords) large
        batches run as a copy / kernel / copy pipeline; `out` is allocated (pageable) when not given."""
        tick_in = np.ascontiguousarray(tick_in, dtype=TICK_IN).reshape(-1)
        if out is None:
            out = np.zeros(tick_in.shape[0], dtype=TICK_OUT)
        assert out.dtype == TICK_OUT and out.flags.c_contiguous and out.shape == (tick_in.shape[0],)
        self._check(self._lib.ismpc_solve_batch(self._h, tick_in.shape[0], tick_in.ctypes.data_as(C.c_void_p),
                                                out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- batch, device buffers (raw pointers; torch tensors via .data_ptr()) ----
    def solve_batch_device(self, batch, in_ptr, out_ptr, u_traj_ptr=None, stream=None):
        self._check(self._lib.ismpc_solve_batch_device(self._h, int(batch), C.c_void_p(in_ptr), C.c_void_p(out_ptr),
                                                       C.c_void_p(u_traj_ptr) if u_traj_ptr else None,
                                                       C.c_void_p(stream) if stream else None))

    def rollout_device(self, batch, state_ptr, first_frame, ticks, traj_ptr=None, stream=None):
        self._check(self._lib.ismpc_rollout_device(self._h, int(batch), C.c_void_p(state_ptr), int(first_frame), int(ticks),
                                                   C.c_void_p(traj_ptr) if traj_ptr else None,
                                                   C.c_void_p(stream) if stream else None))

    def reserve(self, max_batch):
        """Sizes the handle's per-launch scratch now (ismpc_reserve): callers that capture launches into a hipGraph call this first."""
        self._check(self._lib.ismpc_reserve(self._h, int(max_batch)))

    def fallback_counters(self):
        """ismpc_fallback_counters: (deferred-list entries, fallback workgroups done, parked instances, resume workgroups done); all 0 between calls."""
        a = np.zeros(4, dtype=np.int32)
        self._check(self._lib.ismpc_fallback_counters(self._h, a.ctypes.data_as(C.c_void_p)))
        return tuple(int(v) for v in a)

    def set_timing(self, enabled=True):
        self._check(self._lib.ismpc_set_timing(self._h, 1 if enabled else 0))

    def last_kernel_ms(self):
        return float(self._lib.ismpc_last_kernel_ms(self._h))

    def midpoint(self):
        n = self._lib.ismpc_midpoint_rows(self._h)
        m = np.zeros((n, 3))
        self._check(self._lib.ismpc_get_midpoint(self._h, m.ctypes.data_as(C.c_void_p), n))
        return m

    # ---- torch conveniences (device memory and streams are torch's; the compute is not) ----
    def solve_batch_torch(self, tick_in_u8, out_u8=None, u_traj=None):
        """tick_in_u8: CUDA uint8 tensor [batch, 72]; returns CUDA uint8 tensor [batch, 80]."""
        import torch
        b = tick_in_u8.shape[0]
        assert tick_in_u8.is_cuda and tick_in_u8.dtype == torch.uint8 and tick_in_u8.shape[1] == 72 and tick_in_u8.is_contiguous()
        if out_u8 is None:
            out_u8 = torch.empty((b, 80), dtype=torch.uint8, device=tick_in_u8.device)
        stream = torch.cuda.current_stream(tick_in_u8.device).cuda_stream
        self.solve_batch_device(b, tick_in_u8.data_ptr(), out_u8.data_ptr(),
                                u_traj.data_ptr() if u_traj is not None else None, stream)
        return out_u8

    def rollout_torch(self, state_u8, first_frame, ticks, want_traj=True):
        import torch
        b = state_u8.shape[0]
        assert state_u8.is_cuda and state_u8.dtype == torch.uint8 and state_u8.shape[1] == 72 and state_u8.is_contiguous()
        traj = torch.empty((ticks, b, 80), dtype=torch.uint8, device=state_u8.device) if want_traj else None
        stream = torch.cuda.current_stream(state_u8.device).cuda_stream
        self.rollout_device(b, state_u8.data_ptr(), first_frame, ticks, traj.data_ptr() if want_traj else None, stream)
        return traj


def to_device(records, device="cuda:0"):
    """numpy structured records -> CUDA uint8 tensor [batch, itemsize]."""
    import torch
    a = np.ascontiguousarray(records).reshape(-1)
    return torch.from_numpy(a.view(np.uint8).reshape(a.shape[0], a.dtype.itemsize).copy()).to(device)


def from_device(t, dtype):
    a = t.detach().cpu().numpy()
    return np.ascontiguousarray(a).view(dtype).reshape(a.shape[:-1])
