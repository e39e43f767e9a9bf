"""Multi-GPU groups over the C ABI of include/ismpc_group.h: a batch of independent gait instances cut into contiguous
shards, one per GPU, and the path's one collective -- the all-gather of the 80-byte output records -- issued by the native
library on RCCL.  Nothing here computes or communicates in Python; torch only lends device memory.

    Group(plan, params, devices=[0, 1, ...])                      one process drives n devices   (ismpc_group_create)
    Group.from_rank(plan, params, device, uid, rank, world)       one process per GPU             (ismpc_group_create_rank)
    uid = unique_id()  on rank 0, distributed by the caller (bench.py: torch.distributed over gloo)
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import TICK_IN, TICK_OUT

EXPORTS_GROUP = ["ismpc_shard_range", "ismpc_group_unique_id", "ismpc_group_create", "ismpc_group_create_rank", "ismpc_group_destroy",
                 "ismpc_group_world", "ismpc_group_local", "ismpc_group_rank", "ismpc_group_handle", "ismpc_group_solve_batch",
                 "ismpc_group_step_device", "ismpc_group_result_device", "ismpc_group_wait_on", "ismpc_group_sync", "ismpc_group_reserve", "ismpc_group_order_after",
                 "ismpc_a_group_create", "ismpc_a_group_create_rank", "ismpc_a_group_destroy", "ismpc_a_group_world", "ismpc_a_group_local",
                 "ismpc_a_group_rank", "ismpc_a_group_handle", "ismpc_a_group_add_plan", "ismpc_a_group_set_precision", "ismpc_a_group_tick_batch",
                 "ismpc_a_group_step_device", "ismpc_a_group_result_device", "ismpc_a_group_wait_on", "ismpc_a_group_sync", "ismpc_a_group_reserve", "ismpc_a_group_order_after",
                 "ismpc_group_last_error", "ismpc_group_rccl_version"]
UNIQUE_ID_BYTES = 128
_bound = False


class GroupError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ismpc group error {code}: {msg}")
        self.code = code


def _l():
    global _bound
    lib = _lib.load()
    if not _bound:
        vp, ci = C.c_void_p, C.c_int
        pi = C.POINTER(C.c_int)
        lib.ismpc_shard_range.argtypes = [ci, ci, ci, pi, pi]; lib.ismpc_shard_range.restype = ci
        lib.ismpc_group_unique_id.argtypes = [vp]; lib.ismpc_group_unique_id.restype = ci
        lib.ismpc_group_create.argtypes = [vp, vp, ci, pi, ci, C.POINTER(vp)]; lib.ismpc_group_create.restype = ci
        lib.ismpc_group_create_rank.argtypes = [vp, vp, ci, ci, vp, ci, ci, C.POINTER(vp)]; lib.ismpc_group_create_rank.restype = ci
        lib.ismpc_a_group_create.argtypes = [vp, vp, pi, ci, C.POINTER(vp)]; lib.ismpc_a_group_create.restype = ci
        lib.ismpc_a_group_create_rank.argtypes = [vp, vp, ci, vp, ci, ci, C.POINTER(vp)]; lib.ismpc_a_group_create_rank.restype = ci
        for pre in ("ismpc_group_", "ismpc_a_group_"):
            getattr(lib, pre + "destroy").argtypes = [vp]; getattr(lib, pre + "destroy").restype = None
            for f in ("world", "local", "sync"):
                getattr(lib, pre + f).argtypes = [vp]; getattr(lib, pre + f).restype = ci
            getattr(lib, pre + "rank").argtypes = [vp, ci]; getattr(lib, pre + "rank").restype = ci
            getattr(lib, pre + "handle").argtypes = [vp, ci]; getattr(lib, pre + "handle").restype = vp
            getattr(lib, pre + "result_device").argtypes = [vp, ci, ci, C.POINTER(vp)]; getattr(lib, pre + "result_device").restype = ci
            getattr(lib, pre + "wait_on").argtypes = [vp, ci, ci, vp]; getattr(lib, pre + "wait_on").restype = ci
            getattr(lib, pre + "order_after").argtypes = [vp, ci, vp]; getattr(lib, pre + "order_after").restype = ci
            getattr(lib, pre + "reserve").argtypes = [vp, ci]; getattr(lib, pre + "reserve").restype = ci
        lib.ismpc_group_solve_batch.argtypes = [vp, ci, vp, vp]; lib.ismpc_group_solve_batch.restype = ci
        lib.ismpc_group_step_device.argtypes = [vp, ci, C.POINTER(vp), ci]; lib.ismpc_group_step_device.restype = ci
        lib.ismpc_a_group_add_plan.argtypes = [vp, vp]; lib.ismpc_a_group_add_plan.restype = ci
        lib.ismpc_a_group_set_precision.argtypes = [vp, ci]; lib.ismpc_a_group_set_precision.restype = ci
        lib.ismpc_a_group_tick_batch.argtypes = [vp, ci, vp, vp, vp, vp]; lib.ismpc_a_group_tick_batch.restype = ci
        lib.ismpc_a_group_step_device.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), ci]; lib.ismpc_a_group_step_device.restype = ci
        lib.ismpc_group_last_error.argtypes = []; lib.ismpc_group_last_error.restype = C.c_char_p
        lib.ismpc_group_rccl_version.argtypes = []; lib.ismpc_group_rccl_version.restype = ci
        _bound = True
    return lib


def _check(rc):
    if rc < 0:
        raise GroupError(rc, _l().ismpc_group_last_error().decode())
    return rc


def shard_range(batch, rank, world):
    """ismpc_shard_range: contiguous shard [first, first + count) of `rank`; the first batch % world ranks hold one more."""
    f, c = C.c_int(), C.c_int()
    _check(_l().ismpc_shard_range(int(batch), int(rank), int(world), C.byref(f), C.byref(c)))
    return f.value, c.value


def unique_id():
    """128 opaque bytes naming a new RCCL communicator (rank 0 makes them, every rank of the group passes the same bytes)."""
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    _check(_l().ismpc_group_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


def rccl_version():
    return int(_l().ismpc_group_rccl_version())


def _ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p)) if p else None for p in ptrs])


class _DeviceBytes:
    """Device memory owned by the native library, described to torch through the CUDA array interface."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class _GroupBase:
    _pre = "ismpc_group_"

    def _f(self, name):
        return getattr(self._lib, self._pre + name)

    @property
    def world(self):
        """Ranks of the communicator as RCCL itself reports them (ncclCommCount)."""
        return _check(self._f("world")(self._g))

    @property
    def n_local(self):
        return _check(self._f("local")(self._g))

    def rank(self, local=0):
        return _check(self._f("rank")(self._g, int(local)))

    def shard(self, batch, local=0):
        return shard_range(batch, self.rank(local), self.world)

    def reserve(self, max_batch):
        _check(self._f("reserve")(self._g, int(max_batch)))

    def sync(self):
        _check(self._f("sync")(self._g))

    def wait_on(self, local, buf, stream):
        _check(self._f("wait_on")(self._g, int(local), int(buf), C.c_void_p(int(stream)) if stream else None))

    def order_after(self, local, stream):
        """The group's launch stream of local device `local` waits for what is enqueued so far on the caller's stream."""
        _check(self._f("order_after")(self._g, int(local), C.c_void_p(int(stream)) if stream else None))

    def result_ptr(self, local, buf):
        p = C.c_void_p()
        _check(self._f("result_device")(self._g, int(local), int(buf), C.byref(p)))
        return p.value

    def result_torch(self, batch, local=0, buf=0):
        """The gathered buffer of local device `local` as a uint8 tensor [batch, 80]: a zero-copy VIEW of the group's memory (valid until the
        step after next overwrites the buffer).  The current stream of that device is made to wait for the collective first."""
        import torch
        dev = torch.device("cuda", self.devices[local])
        self.wait_on(local, buf, torch.cuda.current_stream(dev).cuda_stream)
        view = _DeviceBytes(self.result_ptr(local, buf), int(batch) * 80)
        return torch.as_tensor(view, device=dev).view(int(batch), 80)

    def close(self):
        if getattr(self, "_g", None):
            self._f("destroy")(self._g); self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group(_GroupBase):
    """Formulation B (MPCSolver::solve) on several GPUs: ismpc_group_*."""

    def __init__(self, ftsp_and_timings, params, devices=(0,), _rank=None):
        self._lib = _l()
        self.params = params
        self.ftsp = np.ascontiguousarray(ftsp_and_timings, dtype=np.float64)
        if self.ftsp.ndim != 2 or self.ftsp.shape[1] != 4:
            raise ValueError("ftsp_and_timings must be rows x 4")
        g = C.c_void_p()
        if _rank is None:
            self.devices = [int(d) for d in devices]
            arr = (C.c_int * len(self.devices))(*self.devices)
            _check(self._lib.ismpc_group_create(C.cast(C.byref(params), C.c_void_p), self.ftsp.ctypes.data_as(C.c_void_p), self.ftsp.shape[0],
                                                arr, len(self.devices), C.byref(g)))
        else:
            device, uid, rank, world = _rank
            if len(uid) != UNIQUE_ID_BYTES:
                raise ValueError("the unique id is 128 bytes")
            self.devices = [int(device)]
            ub = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(uid)
            _check(self._lib.ismpc_group_create_rank(C.cast(C.byref(params), C.c_void_p), self.ftsp.ctypes.data_as(C.c_void_p), self.ftsp.shape[0],
                                                     int(device), C.cast(ub, C.c_void_p), int(rank), int(world), C.byref(g)))
        self._g = g

    @classmethod
    def from_rank(cls, ftsp_and_timings, params, device, uid, rank, world):
        return cls(ftsp_and_timings, params, _rank=(device, uid, rank, world))

    def solve_batch(self, tick_in, out=None):
        """Host records in, ALL host records out (ismpc_group_solve_batch)."""
        tick_in = np.ascontiguousarray(tick_in, dtype=TICK_IN).reshape(-1)
        if out is None:
            out = np.zeros(tick_in.shape[0], dtype=TICK_OUT)
        assert out.dtype == TICK_OUT and out.flags.c_contiguous and out.shape == tick_in.shape
        _check(self._lib.ismpc_group_solve_batch(self._g, tick_in.shape[0], tick_in.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def step_device(self, batch, shard_ptrs, buf):
        """One asynchronous step: shard_ptrs[l] = device pointer of local device l's shard of ismpc_tick_in records."""
        _check(self._lib.ismpc_group_step_device(self._g, int(batch), _ptr_array(shard_ptrs), int(buf)))


class GroupA(_GroupBase):
    """Formulation A (the MATLAB generators' tick) on several GPUs: ismpc_a_group_*."""
    _pre = "ismpc_a_group_"

    def __init__(self, params, center, devices=(0,), _rank=None):
        self._lib = _l()
        self.params = params
        self.center = np.ascontiguousarray(center, dtype=np.float64)
        g = C.c_void_p()
        if _rank is None:
            self.devices = [int(d) for d in devices]
            arr = (C.c_int * len(self.devices))(*self.devices)
            _check(self._lib.ismpc_a_group_create(C.cast(C.byref(params), C.c_void_p), self.center.ctypes.data_as(C.c_void_p), arr, len(self.devices), C.byref(g)))
        else:
            device, uid, rank, world = _rank
            self.devices = [int(device)]
            ub = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(uid)
            _check(self._lib.ismpc_a_group_create_rank(C.cast(C.byref(params), C.c_void_p), self.center.ctypes.data_as(C.c_void_p), int(device),
                                                       C.cast(ub, C.c_void_p), int(rank), int(world), C.byref(g)))
        self._g = g

    @classmethod
    def from_rank(cls, params, center, device, uid, rank, world):
        return cls(params, center, _rank=(device, uid, rank, world))

    def add_plan(self, center):
        c = np.ascontiguousarray(center, dtype=np.float64)
        return _check(self._lib.ismpc_a_group_add_plan(self._g, c.ctypes.data_as(C.c_void_p)))

    def set_precision(self, fp32):
        _check(self._lib.ismpc_a_group_set_precision(self._g, 1 if fp32 else 0))

    def tick_batch(self, state, inst=None, push=None):
        """Host records: `state` (STATE_A, updated in place for this process's shards), optional per-instance records and pushes; returns all
        output records (ismpc_a_group_tick_batch)."""
        from .formulation_a import STATE_A, OUT_A, INST_A
        assert state.dtype == STATE_A and state.flags.c_contiguous
        n = state.shape[0]
        out = np.zeros(n, dtype=OUT_A)
        ip = None
        if inst is not None:
            inst = np.ascontiguousarray(inst, dtype=INST_A); assert inst.shape == (n,)
            ip = inst.ctypes.data_as(C.c_void_p)
        pp = None
        if push is not None:
            push = np.ascontiguousarray(push, dtype=np.float64); assert push.shape == (n, 2)
            pp = push.ctypes.data_as(C.c_void_p)
        _check(self._lib.ismpc_a_group_tick_batch(self._g, n, state.ctypes.data_as(C.c_void_p), ip, pp, out.ctypes.data_as(C.c_void_p)))
        return out

    def step_device(self, batch, state_ptrs, inst_ptrs, push_ptrs, buf):
        _check(self._lib.ismpc_a_group_step_device(self._g, int(batch), _ptr_array(state_ptrs), _ptr_array(inst_ptrs) if inst_ptrs else None,
                                                   _ptr_array(push_ptrs) if push_ptrs else None, int(buf)))
