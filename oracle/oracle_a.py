"""TEST INFRASTRUCTURE ONLY -- ctypes door to the Formulation-A oracle (oracle/ismpc_oracle_a.c):
the MATLAB ISMPC generators (trotting/*.m, walking/*.m) restated in C.  Same QP backends as oracle.py."""
import ctypes as C

import numpy as np

from . import oracle as O


class Gait(C.Structure):
    _fields_ = [("gait", C.c_int), ("n_gait", C.c_int), ("disp_A", C.c_double), ("phi", C.c_double),
                ("disp_B", C.c_double), ("disp_C", C.c_double), ("disp_i", C.c_double), ("disp_o", C.c_double),
                ("disp_forw", C.c_double)]


class ParamsA(C.Structure):
    _fields_ = [("C", C.c_int), ("P", C.c_int), ("F", C.c_int), ("step", C.c_int), ("ds", C.c_int), ("n_gait", C.c_int),
                ("dt", C.c_double), ("height", C.c_double), ("grav", C.c_double), ("w", C.c_double), ("Qf", C.c_double),
                ("disp_forw", C.c_double), ("disp_forw_dummy", C.c_double), ("disp_L", C.c_double)]


STATE_A = np.dtype([("x", "<f8"), ("xd", "<f8"), ("xz", "<f8"), ("y", "<f8"), ("yd", "<f8"), ("yz", "<f8"),
                    ("cur_x", "<f8"), ("cur_y", "<f8"), ("pred_x", "<f8"), ("pred_y", "<f8"),
                    ("fc", "<i4"), ("j", "<i4")], align=False)
TICK_A = np.dtype([("com_before", "<f8", 2), ("vel_after", "<f8", 2), ("u0", "<f8", 2), ("f0", "<f8", 2),
                   ("rv", "<i4", 2), ("nwsr", "<i4", 2), ("fc", "<i4"), ("stepped", "<i4")], align=False)
assert STATE_A.itemsize == 88 and TICK_A.itemsize == 88

TROT, WALK = 0, 1


def gait(kind, phi, disp_A, n_gait=100):
    """init_quadruped.m:5-37 / init_quadruped2.m:5-37 defaults."""
    return Gait(gait=kind, n_gait=n_gait, disp_A=disp_A, phi=phi, disp_B=0.259394, disp_C=0.88,
                disp_i=0.4, disp_o=0.4, disp_forw=0.5)


def params(kind, C_=None, P=None, F=3, step=None, ds=None, Qf=None, n_gait=100):
    """quad_walk_no_plots.m:15-45,270-271 / quad_as_bip_no_plots.m:15-45,256-257."""
    if kind == WALK:
        d = dict(C=100, P=200, step=50, ds=30, Qf=1e9)
    else:
        d = dict(C=160, P=320, step=80, ds=50, Qf=1e7)
    if C_ is not None: d["C"] = C_
    if P is not None: d["P"] = P
    if step is not None: d["step"] = step
    if ds is not None: d["ds"] = ds
    if Qf is not None: d["Qf"] = Qf
    return ParamsA(C=d["C"], P=d["P"], F=F, step=d["step"], ds=d["ds"], n_gait=n_gait, dt=0.01, height=0.56,
                   grav=9.8, w=0.02, Qf=d["Qf"], disp_forw=0.5, disp_forw_dummy=0.25, disp_L=0.4)


def _lib():
    lib = O.lib()
    if not getattr(lib, "_a_ready", False):
        lib.orc_a_plan.argtypes = [C.POINTER(Gait), C.c_void_p, C.c_void_p]
        lib.orc_a_create.restype = C.c_void_p
        lib.orc_a_create.argtypes = [C.POINTER(ParamsA), C.c_void_p, C.c_double]
        lib.orc_a_destroy.argtypes = [C.c_void_p]
        lib.orc_a_set_qp_backend.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_a_get_state.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_a_set_state.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_a_plan_rows.argtypes = [C.c_void_p]
        lib.orc_a_cl_len.argtypes = [C.c_void_p]
        lib.orc_a_get_plan.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        lib.orc_a_set_plan.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int]
        lib.orc_a_tick.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_a_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_a_axis_data.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        lib.orc_a_enable_feet.argtypes = [C.c_void_p, C.POINTER(Gait), C.c_void_p, C.c_int]
        lib.orc_a_foot_rows.argtypes = [C.c_void_p]
        lib.orc_a_get_foot_plan.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_a_foot_trajectories.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_a_load_shifted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int]
        lib._a_ready = True
    return lib


def plan(g):
    """(foot_plan [rows, 8], center [n_gait, 2]) -- 0-based copies of the 1-based MATLAB arrays."""
    fp = np.zeros((g.n_gait + 2, 8)); ce = np.zeros((g.n_gait + 1, 2))
    used = _lib().orc_a_plan(C.byref(g), fp.ctypes.data_as(C.c_void_p), ce.ctypes.data_as(C.c_void_p))
    return fp[1:used + 1].copy(), ce[1:].copy()


class SimA:
    """One MATLAB run: `init_quadruped*` + the `for j = 1:sim_duration` loop, tick by tick."""

    def __init__(self, g, p, backend="auto"):
        self.g, self.p = g, p
        fp = np.zeros((g.n_gait + 2, 8)); self._center = np.zeros((g.n_gait + 1, 2))
        _lib().orc_a_plan(C.byref(g), fp.ctypes.data_as(C.c_void_p), self._center.ctypes.data_as(C.c_void_p))
        self._h = _lib().orc_a_create(C.byref(p), self._center.ctypes.data_as(C.c_void_p), g.disp_C)
        if backend == "auto":
            backend = "ref" if O.have_ref() else "gi"
        self.backend = backend
        if backend == "ref":
            _lib().orc_a_set_qp_backend(self._h, C.cast(O.ref_lib().qpoases_ref_solve, C.c_void_p))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().orc_a_destroy(self._h); self._h = None

    def run(self, ticks):
        outs = np.zeros(ticks, dtype=TICK_A)
        _lib().orc_a_run(self._h, ticks, outs.ctypes.data_as(C.c_void_p))
        return outs

    def tick(self, push=(0.0, 0.0), want_solution=False):
        out = np.zeros(1, dtype=TICK_A)
        nv = self.p.C + self.p.F
        sx = np.zeros(nv); sy = np.zeros(nv)
        _lib().orc_a_tick(self._h, float(push[0]), float(push[1]), out.ctypes.data_as(C.c_void_p),
                          sx.ctypes.data_as(C.c_void_p), sy.ctypes.data_as(C.c_void_p))
        return (out[0], sx, sy) if want_solution else out[0]

    def enable_feet(self):
        """Turn on the swing-foot re-placement QPs (second quadprog of the scripts) for this run."""
        fp = np.zeros((self.g.n_gait + 2, 8)); ce = np.zeros((self.g.n_gait + 1, 2))
        used = _lib().orc_a_plan(C.byref(self.g), fp.ctypes.data_as(C.c_void_p), ce.ctypes.data_as(C.c_void_p))
        self._fp0 = np.ascontiguousarray(fp[1:used + 1])
        _lib().orc_a_enable_feet(self._h, C.byref(self.g), self._fp0.ctypes.data_as(C.c_void_p), used)

    def foot_plan(self):
        n = _lib().orc_a_foot_rows(self._h)
        fp = np.zeros((n, 8)); _lib().orc_a_get_foot_plan(self._h, fp.ctypes.data_as(C.c_void_p))
        return fp

    def foot_trajectories(self, sim_duration=2000):
        """foot_{fl,fr,rl,rr}_*.txt contents: array [4, rows, 3]."""
        rows = (sim_duration // self.p.step) * self.p.step
        out = np.zeros((4, rows, 3))
        _lib().orc_a_foot_trajectories(self._h, sim_duration, out.ctypes.data_as(C.c_void_p))
        return out

    def axis_data(self, axis):
        """Structured per-axis QP of the NEXT tick: dict(a, b, zlo, zhi, M [C, F+1], klo, khi, pref)."""
        Cn, F = self.p.C, self.p.F
        a = np.zeros(Cn); b = np.zeros(1); zlo = np.zeros(Cn); zhi = np.zeros(Cn); M = np.zeros((Cn, F + 1))
        klo = np.zeros(F); khi = np.zeros(F); pref = np.zeros(F)
        rc = _lib().orc_a_axis_data(self._h, axis, *[x.ctypes.data_as(C.c_void_p) for x in (a, b, zlo, zhi, M, klo, khi, pref)])
        if rc != 0:
            raise RuntimeError("mapping overflow")
        return dict(a=a, b=float(b[0]), zlo=zlo, zhi=zhi, M=M, klo=klo, khi=khi, pref=pref)

    @property
    def state(self):
        st = np.zeros(1, dtype=STATE_A)
        _lib().orc_a_get_state(self._h, st.ctypes.data_as(C.c_void_p))
        return st[0]

    @state.setter
    def state(self, st):
        st = np.ascontiguousarray(st, dtype=STATE_A).reshape(1)
        _lib().orc_a_set_state(self._h, st.ctypes.data_as(C.c_void_p))

    def load_product_state(self, rec):
        """Mid-run situation from one record of the batched generators' per-instance state (x..cur_y, off_x/off_y =
        plan shift, fc, j, rebuilt): fs_plan = base plan + shift, centreline rebuilt (quad_walk_no_plots.m:535-549)."""
        st = np.zeros(1, dtype=STATE_A)
        for k in ("x", "xd", "xz", "y", "yd", "yz", "cur_x", "cur_y", "fc", "j"):
            st[k] = rec[k]
        _lib().orc_a_load_shifted(self._h, st.ctypes.data_as(C.c_void_p), self._center.ctypes.data_as(C.c_void_p),
                                  float(rec["off_x"]), float(rec["off_y"]), int(rec["rebuilt"]))

    def get_plan(self):
        n, m = _lib().orc_a_plan_rows(self._h), _lib().orc_a_cl_len(self._h)
        fsx, fsy, clx, cly = np.zeros(n), np.zeros(n), np.zeros(m), np.zeros(m)
        _lib().orc_a_get_plan(self._h, *[a.ctypes.data_as(C.c_void_p) for a in (fsx, fsy, clx, cly)])
        return fsx, fsy, clx, cly

    def set_plan(self, fsx, fsy, clx, cly):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (fsx, fsy, clx, cly)]
        _lib().orc_a_set_plan(self._h, *[a.ctypes.data_as(C.c_void_p) for a in arrs], len(arrs[2]))
