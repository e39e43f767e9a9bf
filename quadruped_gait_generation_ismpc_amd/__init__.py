"""MI355X-native ISMPC gait-generation hot path (the per-tick MPCSolver loop of
FrancescoScotti/Quadruped_gait_generation_ISMPC) -- host-side mirror of the reference
interface over the C ABI in include/ismpc.h.  The compute path is HIP on gfx950 only."""
from ._lib import Params, TICK_IN, TICK_OUT, NativeLibraryMissing, LIB_PATH, EXPORTS, PinnedRecords  # noqa: F401
from .solver import (MPCSolver, State, WalkState, IsmpcError, default_params, reference_plan,  # noqa: F401
                     to_device, from_device,
                     ST_X_INFEASIBLE, ST_Y_INFEASIBLE, ST_Z_INEQ_ACTIVE, ST_BAD_INDEX,
                     ST_FLIGHT, ST_TICK_SKIPPED, ST_Z_NAN, ST_Z_FAILED, ST_ERROR_MASK)
