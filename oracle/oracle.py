"""TEST INFRASTRUCTURE ONLY -- ctypes door to the CPU oracle.

Loads oracle/libismpc_oracle.so (this repo's plain-C restatement of
MPCSolver.cpp:5-200,204-430) and, when present, oracle/_ref/libqpoases_ref.so
(the reference's own vendored qpOASES 3.2 built in place from /root/reference
by oracle/Makefile) and plugs the latter in as the QP backend.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libismpc_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libqpoases_ref.so")


class Params(C.Structure):
    """Mirror of ismpc_params (include/ismpc.h)."""
    _fields_ = [("N", C.c_int32), ("S", C.c_int32), ("F", C.c_int32), ("M", C.c_int32),
                ("mpc_dt", C.c_double), ("control_dt", C.c_double), ("mass", C.c_double),
                ("g", C.c_double), ("h_des", C.c_double), ("foot_width", C.c_double),
                ("first_step_halfwidth", C.c_double),
                ("q_p", C.c_double), ("q_u", C.c_double), ("q_v", C.c_double),
                ("z_ineq_lo", C.c_double), ("z_ineq_hi", C.c_double),
                ("lambda_gate", C.c_double)]


def default_params(N=100, **kw):
    """parameters.cpp:9-45 and MPCSolver.cpp:253-255,159-160,322."""
    p = Params(N=N, S=35, F=10, M=2, mpc_dt=0.01, control_dt=0.01, mass=50.0, g=9.81,
               h_des=0.69, foot_width=0.09, first_step_halfwidth=1.0,
               q_p=1005000.0, q_u=0.01, q_v=100.0, z_ineq_lo=0.0, z_ineq_hi=10000.0,
               lambda_gate=2.0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


TICK_IN = np.dtype([("com_pos", "<f8", 3), ("com_vel", "<f8", 3), ("simulation_time", "<f8"),
                    ("mpc_iter", "<i4"), ("control_iter", "<i4"), ("footstep_counter", "<i4"),
                    ("reserved", "<i4")], align=False)
TICK_OUT = np.dtype([("com_pos", "<f8", 3), ("com_vel", "<f8", 3), ("u0", "<f8", 3),
                     ("status", "<i4"), ("iters", "<i4")], align=False)
TICK_INFO = np.dtype([("rv", "<i4", 3), ("nwsr", "<i4", 3), ("idx", "<i4"), ("ne_z", "<i4"),
                      ("lambda0", "<f8"), ("beq", "<f8", 2)], align=False)
assert TICK_IN.itemsize == 72 and TICK_OUT.itemsize == 80 and TICK_INFO.itemsize == 56

ST_X_INFEASIBLE, ST_Y_INFEASIBLE, ST_Z_INEQ_ACTIVE, ST_BAD_INDEX = 1, 2, 4, 8
ST_FLIGHT, ST_TICK_SKIPPED, ST_Z_NAN, ST_Z_FAILED = 16, 32, 64, 128
ST_ERROR_MASK = ST_X_INFEASIBLE | ST_Y_INFEASIBLE | ST_BAD_INDEX | ST_Z_FAILED


def reference_plan(rows=40, S=35, F=10, mpc_dt=0.01, control_dt=0.01):
    """ftsp_and_time of Controller.cpp:89-97 (row 0 stays zero)."""
    ftsp = np.zeros((rows, 4))
    for i in range(1, rows):
        ftsp[i, 0] = (i - 1) * 0.2
        ftsp[i, 1] = (-1.0) ** (i - 1) * 0.08
        ftsp[i, 2] = 0.0
        ftsp[i, 3] = (mpc_dt / control_dt) * (S + F) * i
    return ftsp


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(ORACLE_SO) or \
            any(os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(ORACLE_SO)
                for f in os.listdir(HERE) if f.startswith("ismpc_oracle") and f.endswith(".c")):
        subprocess.check_call(["make", "-C", HERE, "libismpc_oracle.so"], stdout=subprocess.DEVNULL)
    if not os.path.exists(REF_SO) and os.path.isdir("/root/reference/AMR_code_DART/qpOASES"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_QP_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int))
_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(ORACLE_SO)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_create.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int]
        _lib.orc_destroy.argtypes = [C.c_void_p]
        _lib.orc_set_qp_backend.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_midpoint_rows.argtypes = [C.c_void_p]
        _lib.orc_midpoint.restype = C.c_void_p
        _lib.orc_midpoint.argtypes = [C.c_void_p]
        _lib.orc_Hz.restype = C.c_void_p
        _lib.orc_Hz.argtypes = [C.c_void_p]
        _lib.orc_solve_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_qp_gi.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.POINTER(C.c_int)]
    return _lib


def ref_lib():
    """The reference's qpOASES (oracle/_ref), or None when it was never built."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        _ref = C.CDLL(REF_SO)
        _ref.qpoases_ref_solve.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.POINTER(C.c_int)]
        _ref.qpoases_ref_solve.restype = C.c_int
    return _ref


def have_ref():
    return ref_lib() is not None


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def solve_qp(H, g, A, lbA, ubA, backend="gi", nwsr=300):
    """Dense QP in the reference's solveQP convention (utils.cpp:89-139)."""
    H = np.ascontiguousarray(H, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(-1, H.shape[0])
    lbA = np.ascontiguousarray(lbA, dtype=np.float64); ubA = np.ascontiguousarray(ubA, dtype=np.float64)
    x = np.zeros(H.shape[0]); n = C.c_int(nwsr)
    if backend == "ref":
        fn = ref_lib().qpoases_ref_solve
    else:
        fn = lib().orc_qp_gi
    rv = fn(H.shape[0], A.shape[0], _ptr(H), _ptr(g), _ptr(A), _ptr(lbA), _ptr(ubA), _ptr(x), C.byref(n))
    return x, rv, n.value


class Oracle:
    """CPU restatement of MPCSolver (ctor + solve), one instance at a time."""

    def __init__(self, params=None, ftsp=None, backend="auto"):
        self.params = params if params is not None else default_params()
        self.ftsp = np.ascontiguousarray(ftsp if ftsp is not None else
                                         reference_plan(S=self.params.S, F=self.params.F,
                                                        mpc_dt=self.params.mpc_dt,
                                                        control_dt=self.params.control_dt), dtype=np.float64)
        self._h = lib().orc_create(C.byref(self.params), _ptr(self.ftsp), self.ftsp.shape[0])
        if not self._h:
            raise ValueError("orc_create rejected the parameters")
        if backend == "auto":
            backend = "ref" if have_ref() else "gi"
        self.backend = backend
        if backend == "ref":
            if not have_ref():
                raise RuntimeError("oracle/_ref/libqpoases_ref.so is not built")
            lib().orc_set_qp_backend(self._h, C.cast(ref_lib().qpoases_ref_solve, C.c_void_p))
        else:
            lib().orc_set_qp_backend(self._h, None)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h); self._h = None

    @property
    def N(self):
        return self.params.N

    def midpoint(self):
        n = lib().orc_midpoint_rows(self._h)
        p = lib().orc_midpoint(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n, 3)).copy()

    def Hz(self):
        p = lib().orc_Hz(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(self.N, self.N)).copy()

    def solve(self, tick_in, want_traj=False):
        tick_in = np.ascontiguousarray(tick_in, dtype=TICK_IN).reshape(-1)
        b = tick_in.shape[0]
        out = np.zeros(b, dtype=TICK_OUT); info = np.zeros(b, dtype=TICK_INFO)
        traj = np.zeros((b, 3, self.N)) if want_traj else None
        lib().orc_solve_batch(self._h, b, _ptr(tick_in), _ptr(out), _ptr(traj) if want_traj else None, _ptr(info))
        return (out, info, traj) if want_traj else (out, info)

    def rollout(self, state, first_frame, ticks):
        """Closed loop (Controller.cpp:297-310,346-348,503-504). Returns (outs, ins, infos, final_state)."""
        st = np.ascontiguousarray(state, dtype=TICK_IN).reshape(1).copy()
        outs = np.zeros(ticks, dtype=TICK_OUT); ins = np.zeros(ticks, dtype=TICK_IN)
        infos = np.zeros(ticks, dtype=TICK_INFO)
        lib().orc_rollout(self._h, _ptr(st), first_frame, ticks, _ptr(outs), _ptr(ins), _ptr(infos))
        return outs, ins, infos, st


def initial_state(h_des=0.69):
    """State the reference starts from: desired.comPos = (x, y, comTargetHeight), Controller.cpp:110;
    WalkState of Controller.cpp:65-68 (simulationTime is never initialised there: 0)."""
    st = np.zeros(1, dtype=TICK_IN)
    st["com_pos"][0] = (0.0, 0.0, h_des)
    return st
