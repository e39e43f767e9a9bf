"""Host-side mirror of the reference's MATLAB gait generators ("Formulation A") over the C ABI of
include/ismpc_a.h: plan generators (init_quadruped*.m) and the per-tick ISMPC QP with footstep adaptation
(quad_walk_no_plots.m / quad_as_bip_no_plots.m loop body).  All compute is HIP; nothing is solved in Python."""
import ctypes as C
import os

import numpy as np

from . import _lib

TROT, WALK = 0, 1
ST_X_INFEASIBLE, ST_Y_INFEASIBLE, ST_OVERFLOW, ST_BAD_INDEX, ST_ITER_LIMIT, ST_UNVERIFIED = 1, 2, 4, 8, 16, 32


class GaitA(C.Structure):
    _fields_ = [("gait", C.c_int32), ("n_gait", C.c_int32), ("disp_A", C.c_double), ("phi", C.c_double),
                ("disp_B", C.c_double), ("disp_C", C.c_double), ("disp_i", C.c_double), ("disp_o", C.c_double),
                ("disp_forw", C.c_double)]


class ParamsA(C.Structure):
    _fields_ = [("C", C.c_int32), ("P", C.c_int32), ("F", C.c_int32), ("step", C.c_int32), ("ds", C.c_int32),
                ("n_gait", C.c_int32), ("dt", C.c_double), ("height", C.c_double), ("grav", C.c_double),
                ("w", C.c_double), ("Qf", C.c_double), ("disp_forw", C.c_double), ("disp_forw_dummy", C.c_double),
                ("disp_L", C.c_double)]


STATE_A = np.dtype([("x", "<f8"), ("xd", "<f8"), ("xz", "<f8"), ("y", "<f8"), ("yd", "<f8"), ("yz", "<f8"),
                    ("cur_x", "<f8"), ("cur_y", "<f8"), ("off_x", "<f8"), ("off_y", "<f8"),
                    ("fc", "<i4"), ("j", "<i4"), ("rebuilt", "<i4"), ("reserved", "<i4")], align=False)
OUT_A = np.dtype([("com_before", "<f8", 2), ("vel_after", "<f8", 2), ("u0", "<f8", 2), ("f0", "<f8", 2),
                  ("status", "<i4"), ("iters_x", "<i4"), ("iters_y", "<i4"), ("active", "<i4")], align=False)
INST_A = np.dtype([("height", "<f8"), ("Qf", "<f8"), ("step", "<i4"), ("ds", "<i4"), ("F", "<i4"), ("plan", "<i4")], align=False)
assert STATE_A.itemsize == 96 and OUT_A.itemsize == 80 and INST_A.itemsize == 32

EXPORTS_A = ["ismpc_a_params_default", "ismpc_a_gait_default", "ismpc_a_plan", "ismpc_a_create", "ismpc_a_destroy",
             "ismpc_a_initial_state", "ismpc_a_tick_batch_device", "ismpc_a_rollout_device", "ismpc_a_last_error",
             "ismpc_a_feet_rows", "ismpc_a_feet_init_device", "ismpc_a_tick_feet_batch_device",
             "ismpc_a_rollout_feet_device", "ismpc_a_foot_trajectories", "ismpc_a_write_trajectory_txt",
             "ismpc_a_add_plan", "ismpc_a_tick_batch_inst_device", "ismpc_a_rollout_inst_device", "ismpc_a_set_warm_history", "ismpc_a_reserve", "ismpc_a_set_precision",
             "ismpc_a_last_deferred", "ismpc_a_feet_init_inst_device", "ismpc_a_tick_feet_batch_inst_device",
             "ismpc_a_rollout_feet_inst_device"]
HAVE_F32 = True        # the QP solve also exists in fp32 (GaitGenerator(..., precision="f32"))
FEET_PAD = 8

_bound = False


def _l():
    global _bound
    lib = _lib.load()
    if not _bound:
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        lib.ismpc_a_params_default.argtypes = [ci, C.POINTER(ParamsA)]; lib.ismpc_a_params_default.restype = None
        lib.ismpc_a_gait_default.argtypes = [ci, cd, cd, C.POINTER(GaitA)]; lib.ismpc_a_gait_default.restype = None
        lib.ismpc_a_plan.argtypes = [C.POINTER(GaitA), vp, vp]; lib.ismpc_a_plan.restype = ci
        lib.ismpc_a_create.argtypes = [C.POINTER(ParamsA), vp, ci, C.POINTER(vp)]; lib.ismpc_a_create.restype = ci
        lib.ismpc_a_destroy.argtypes = [vp]; lib.ismpc_a_destroy.restype = None
        lib.ismpc_a_initial_state.argtypes = [vp, cd, vp]; lib.ismpc_a_initial_state.restype = ci
        lib.ismpc_a_tick_batch_device.argtypes = [vp, ci, vp, vp, vp, vp]; lib.ismpc_a_tick_batch_device.restype = ci
        lib.ismpc_a_rollout_device.argtypes = [vp, ci, vp, ci, vp, vp]; lib.ismpc_a_rollout_device.restype = ci
        lib.ismpc_a_last_error.argtypes = []; lib.ismpc_a_last_error.restype = C.c_char_p
        lib.ismpc_a_feet_rows.argtypes = [vp]; lib.ismpc_a_feet_rows.restype = ci
        lib.ismpc_a_feet_init_device.argtypes = [vp, C.POINTER(GaitA), vp, ci, ci, vp, vp]; lib.ismpc_a_feet_init_device.restype = ci
        lib.ismpc_a_tick_feet_batch_device.argtypes = [vp, ci, vp, vp, vp, vp, vp]; lib.ismpc_a_tick_feet_batch_device.restype = ci
        lib.ismpc_a_rollout_feet_device.argtypes = [vp, ci, vp, ci, vp, vp, vp]; lib.ismpc_a_rollout_feet_device.restype = ci
        lib.ismpc_a_foot_trajectories.argtypes = [C.POINTER(GaitA), ci, vp, ci, ci, vp]; lib.ismpc_a_foot_trajectories.restype = ci
        lib.ismpc_a_write_trajectory_txt.argtypes = [C.c_char_p, vp, ci]; lib.ismpc_a_write_trajectory_txt.restype = ci
        lib.ismpc_a_set_warm_history.argtypes = [vp, ci]; lib.ismpc_a_set_warm_history.restype = ci
        lib.ismpc_a_add_plan.argtypes = [vp, vp]; lib.ismpc_a_add_plan.restype = ci
        lib.ismpc_a_tick_batch_inst_device.argtypes = [vp, ci, vp, vp, vp, vp, vp]; lib.ismpc_a_tick_batch_inst_device.restype = ci
        lib.ismpc_a_rollout_inst_device.argtypes = [vp, ci, vp, vp, ci, vp, vp]; lib.ismpc_a_rollout_inst_device.restype = ci
        lib.ismpc_a_reserve.argtypes = [vp, ci]; lib.ismpc_a_reserve.restype = ci
        lib.ismpc_a_set_precision.argtypes = [vp, ci]; lib.ismpc_a_set_precision.restype = ci
        lib.ismpc_a_last_deferred.argtypes = [vp]; lib.ismpc_a_last_deferred.restype = ci
        lib.ismpc_a_feet_init_inst_device.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp]; lib.ismpc_a_feet_init_inst_device.restype = ci
        lib.ismpc_a_tick_feet_batch_inst_device.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]; lib.ismpc_a_tick_feet_batch_inst_device.restype = ci
        lib.ismpc_a_rollout_feet_inst_device.argtypes = [vp, ci, vp, vp, ci, vp, vp, vp]; lib.ismpc_a_rollout_feet_inst_device.restype = ci
        _bound = True
    return lib


class IsmpcAError(RuntimeError):
    pass


def default_params(gait, **over):
    p = ParamsA()
    _l().ismpc_a_params_default(int(gait), C.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def default_gait(gait, phi, disp_A):
    g = GaitA()
    _l().ismpc_a_gait_default(int(gait), float(phi), float(disp_A), C.byref(g))
    return g


def plan(g):
    """init_quadruped.m / init_quadruped2.m: (foot_plan [rows, 8], center = fs_plan [n_gait, 2])."""
    fp = np.zeros((g.n_gait + 1, 8)); ce = np.zeros((g.n_gait, 2))
    used = _l().ismpc_a_plan(C.byref(g), fp.ctypes.data_as(C.c_void_p), ce.ctypes.data_as(C.c_void_p))
    if used < 0:
        raise IsmpcAError(_l().ismpc_a_last_error().decode())
    return fp[:used].copy(), ce


def foot_trajectories(g, step, foot_plan, sim_duration=2000):
    """Host: the four foot files (fl, fr, rl, rr) of one instance, [4, rows, 3]  (quad_*_no_plots.m foot writers)."""
    fp = np.ascontiguousarray(foot_plan, dtype=np.float64)
    n = (sim_duration // step) * step
    dst = np.zeros((4, n, 3))
    rc = _l().ismpc_a_foot_trajectories(C.byref(g), int(step), fp.ctypes.data_as(C.c_void_p), fp.shape[0], int(sim_duration),
                                        dst.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise IsmpcAError(_l().ismpc_a_last_error().decode())
    return dst


def write_trajectory_txt(path, rows3):
    """The text wire format the DART controller reads back (Controller.cpp:147-281): MATLAB's '%d %d %d\n'."""
    a = np.ascontiguousarray(rows3, dtype=np.float64).reshape(-1, 3)
    rc = _l().ismpc_a_write_trajectory_txt(os.fsencode(path), a.ctypes.data_as(C.c_void_p), a.shape[0])
    if rc != 0:
        raise IsmpcAError(_l().ismpc_a_last_error().decode())


class GaitGenerator:
    """The MATLAB loop `for j = 1:sim_duration` (quad_walk_no_plots.m:127 / quad_as_bip_no_plots.m:116),
    batched: every instance carries its own (state, footstep counter, plan shift)."""

    def __init__(self, params, center, device=0, precision="f64"):
        """precision: arithmetic type of the QP solve, "f64" (default) or "f32" (ismpc_a_set_precision); the state, the
        right-hand sides of the QP and the LIP update are fp64 either way."""
        if precision not in ("f64", "f32"):
            raise ValueError("precision must be 'f64' or 'f32'")
        self.precision = precision
        self.params = params
        self.center = np.ascontiguousarray(center, dtype=np.float64)
        h = C.c_void_p()
        rc = _l().ismpc_a_create(C.byref(params), self.center.ctypes.data_as(C.c_void_p), int(device), C.byref(h))
        if rc != 0:
            raise IsmpcAError(f"ismpc_a_create: {rc}: {_l().ismpc_a_last_error().decode()}")
        self._h = h
        if _l().ismpc_a_set_precision(self._h, 1 if precision == "f32" else 0) != 0:
            msg = _l().ismpc_a_last_error().decode(); self.close()
            raise IsmpcAError(msg)

    def close(self):
        if getattr(self, "_h", None) and _l is not None:
            _l().ismpc_a_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def initial_state(self, disp_C=0.88, batch=1):
        st = np.zeros(1, dtype=STATE_A)
        rc = _l().ismpc_a_initial_state(self._h, float(disp_C), st.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return np.repeat(st, batch)

    def tick_device(self, batch, state_ptr, push_ptr=None, out_ptr=None, stream=None):
        rc = _l().ismpc_a_tick_batch_device(self._h, int(batch), C.c_void_p(state_ptr),
                                            C.c_void_p(push_ptr) if push_ptr else None,
                                            C.c_void_p(out_ptr) if out_ptr else None,
                                            C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())

    def rollout_device(self, batch, state_ptr, ticks, traj_ptr=None, stream=None):
        rc = _l().ismpc_a_rollout_device(self._h, int(batch), C.c_void_p(state_ptr), int(ticks),
                                         C.c_void_p(traj_ptr) if traj_ptr else None,
                                         C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())

    def feet_init_torch(self, g, foot_plan, batch, device="cuda:0"):
        """Per-instance foot plans on the device: float64 tensor [batch, rows + FEET_PAD, 8]."""
        import torch
        fp = np.ascontiguousarray(foot_plan, dtype=np.float64)
        feet = torch.empty((batch, fp.shape[0] + FEET_PAD, 8), dtype=torch.float64, device=device)
        stream = torch.cuda.current_stream(feet.device).cuda_stream
        rc = _l().ismpc_a_feet_init_device(self._h, C.byref(g), fp.ctypes.data_as(C.c_void_p), fp.shape[0], int(batch),
                                           C.c_void_p(feet.data_ptr()), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return feet

    def rollout_feet_torch(self, state_u8, feet, ticks):
        import torch
        b = state_u8.shape[0]
        traj = torch.empty((ticks, b, 80), dtype=torch.uint8, device=state_u8.device)
        stream = torch.cuda.current_stream(state_u8.device).cuda_stream
        rc = _l().ismpc_a_rollout_feet_device(self._h, b, C.c_void_p(state_u8.data_ptr()), int(ticks), C.c_void_p(traj.data_ptr()),
                                              C.c_void_p(feet.data_ptr()), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return traj

    def last_deferred(self):
        """QPs the last fp32 launch handed to the fp64 re-solve launch behind it (synchronises)."""
        n = _l().ismpc_a_last_deferred(self._h)
        if n < 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return n

    def feet_init_inst_torch(self, gaits, foot_plans, inst_u8):
        """Per-instance gait parameters: one GaitA and one foot_plan per base plan of the handle (same row count); every
        instance starts from the foot plan of its own base plan.  Returns float64 tensor [batch, rows + FEET_PAD, 8]."""
        import torch
        rows = max(np.asarray(f).shape[0] for f in foot_plans)           # the trot generator writes n_gait rows, the walk one n_gait + 1:
        pad = lambda f: np.concatenate([f, np.repeat(f[-1:], rows - f.shape[0], 0)])   # a plan holds its last row beyond its end
        fps = np.ascontiguousarray(np.stack([pad(np.asarray(f, dtype=np.float64)) for f in foot_plans]))
        ga = (GaitA * len(gaits))(*gaits)
        b = inst_u8.shape[0]
        feet = torch.empty((b, fps.shape[1] + FEET_PAD, 8), dtype=torch.float64, device=inst_u8.device)
        stream = torch.cuda.current_stream(inst_u8.device).cuda_stream
        rc = _l().ismpc_a_feet_init_inst_device(self._h, C.cast(ga, C.c_void_p), fps.ctypes.data_as(C.c_void_p), fps.shape[1], len(gaits), b,
                                                C.c_void_p(inst_u8.data_ptr()), C.c_void_p(feet.data_ptr()), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return feet

    def rollout_feet_inst_torch(self, state_u8, inst_u8, feet, ticks):
        import torch
        b = state_u8.shape[0]
        traj = torch.empty((ticks, b, 80), dtype=torch.uint8, device=state_u8.device)
        stream = torch.cuda.current_stream(state_u8.device).cuda_stream
        rc = _l().ismpc_a_rollout_feet_inst_device(self._h, b, C.c_void_p(state_u8.data_ptr()), C.c_void_p(inst_u8.data_ptr()), int(ticks),
                                                   C.c_void_p(traj.data_ptr()), C.c_void_p(feet.data_ptr()), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return traj

    def tick_feet_inst_torch(self, state_u8, inst_u8, feet, push=None):
        import torch
        b = state_u8.shape[0]
        out = torch.empty((b, 80), dtype=torch.uint8, device=state_u8.device)
        stream = torch.cuda.current_stream(state_u8.device).cuda_stream
        rc = _l().ismpc_a_tick_feet_batch_inst_device(self._h, b, C.c_void_p(state_u8.data_ptr()), C.c_void_p(inst_u8.data_ptr()),
                                                      C.c_void_p(push.data_ptr()) if push is not None else None, C.c_void_p(out.data_ptr()),
                                                      C.c_void_p(feet.data_ptr()), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return out

    def set_warm_history(self, enabled=True):
        """Caller-driven tick loops: start every QP from the working set the same instance had in the previous call."""
        if _l().ismpc_a_set_warm_history(self._h, 1 if enabled else 0) != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())

    # per-instance gait parameters (INST_A records): Monte-Carlo / sweep batches
    def add_plan(self, center):
        ce = np.ascontiguousarray(center, dtype=np.float64)
        if ce.shape != self.center.shape:
            raise ValueError("plan shape differs from the handle's")
        k = _l().ismpc_a_add_plan(self._h, ce.ctypes.data_as(C.c_void_p))
        if k < 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return k

    def tick_inst_torch(self, state_u8, inst_u8, push=None):
        import torch
        b = state_u8.shape[0]
        assert state_u8.is_cuda and state_u8.dtype == torch.uint8 and state_u8.shape[1] == 96 and state_u8.is_contiguous()
        assert inst_u8.is_cuda and inst_u8.dtype == torch.uint8 and inst_u8.shape == (b, 32) and inst_u8.is_contiguous()
        out = torch.empty((b, 80), dtype=torch.uint8, device=state_u8.device)
        stream = torch.cuda.current_stream(state_u8.device).cuda_stream
        rc = _l().ismpc_a_tick_batch_inst_device(self._h, b, C.c_void_p(state_u8.data_ptr()), C.c_void_p(inst_u8.data_ptr()),
                                                 C.c_void_p(push.data_ptr()) if push is not None else None,
                                                 C.c_void_p(out.data_ptr()), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return out

    def rollout_inst_torch(self, state_u8, inst_u8, ticks):
        import torch
        b = state_u8.shape[0]
        assert inst_u8.is_cuda and inst_u8.dtype == torch.uint8 and inst_u8.shape == (b, 32) and inst_u8.is_contiguous()
        traj = torch.empty((ticks, b, 80), dtype=torch.uint8, device=state_u8.device)
        stream = torch.cuda.current_stream(state_u8.device).cuda_stream
        rc = _l().ismpc_a_rollout_inst_device(self._h, b, C.c_void_p(state_u8.data_ptr()), C.c_void_p(inst_u8.data_ptr()), int(ticks),
                                              C.c_void_p(traj.data_ptr()), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise IsmpcAError(_l().ismpc_a_last_error().decode())
        return traj

    # torch conveniences
    def tick_torch(self, state_u8, push=None):
        import torch
        b = state_u8.shape[0]
        assert state_u8.is_cuda and state_u8.dtype == torch.uint8 and state_u8.shape[1] == 96 and state_u8.is_contiguous()
        out = torch.empty((b, 80), dtype=torch.uint8, device=state_u8.device)
        stream = torch.cuda.current_stream(state_u8.device).cuda_stream
        self.tick_device(b, state_u8.data_ptr(), push.data_ptr() if push is not None else None, out.data_ptr(), stream)
        return out

    def rollout_torch(self, state_u8, ticks):
        import torch
        b = state_u8.shape[0]
        traj = torch.empty((ticks, b, 80), dtype=torch.uint8, device=state_u8.device)
        stream = torch.cuda.current_stream(state_u8.device).cuda_stream
        self.rollout_device(b, state_u8.data_ptr(), ticks, traj.data_ptr(), stream)
        return traj
