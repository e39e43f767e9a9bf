/*
 * ismpc.h -- C ABI of the MI355X-native ISMPC gait-generation hot path.
 *
 * This is the drop-in boundary for ONE path of the reference
 * (FrancescoScotti/Quadruped_gait_generation_ISMPC): the per-tick
 * `MPCSolver` loop.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference's AMR_code_DART/).
 *
 *   ismpc_create        <->  MPCSolver::MPCSolver(const Eigen::MatrixXd&)      MPCSolver.hpp:18, MPCSolver.cpp:5-200
 *   ismpc_solve_batch*  <->  State MPCSolver::solve(State, WalkState, const&)  MPCSolver.hpp:22, MPCSolver.cpp:204-501
 *                            (which calls solveQP_hpipm_z                      utils.cpp:264-383
 *                             and solveQP_hpipm_xy_piecewiseconstantZMP x2     utils.cpp:385-511)
 *   ismpc_rollout*      <->  the tick bookkeeping around solve()               Controller.cpp:297-310,346-348,503-504
 *   ismpc_destroy       <->  MPCSolver::~MPCSolver()                           MPCSolver.hpp:19
 *   ismpc_params        <->  the compile-time constants of                     parameters.cpp:9-45, MPCSolver.cpp:253-255
 *   ismpc_tick_in/out   <->  the fields of State / WalkState that solve()
 *                            reads and writes                                  types.hpp:7-12,77-81
 *
 * Conventions: plain C types only, no exceptions, int status returns
 * (0 = ok, negative = ISMPC_E_*), caller owns every buffer, a handle is not
 * re-entrant (like the reference's MPCSolver, which keeps member scratch) but
 * distinct handles are independent.  The compute path is HIP on gfx950; there
 * is NO CPU fallback: without a usable GPU every compute entry point returns
 * ISMPC_E_NO_DEVICE and ismpc_last_error() says why.  Every entry point runs on
 * the handle's device and restores the caller's current HIP device before it returns.
 */
#ifndef ISMPC_H
#define ISMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISMPC_ABI_VERSION 1

/* ---- error codes (function return values) ------------------------------ */
#define ISMPC_OK              0
#define ISMPC_E_INVALID      -1   /* bad argument / parameter combination    */
#define ISMPC_E_NO_DEVICE    -2   /* no HIP device, or HIP runtime error     */
#define ISMPC_E_ALLOC        -3   /* host or device allocation failed        */
#define ISMPC_E_NUMERIC      -4   /* vertical Hessian not positive definite  */
#define ISMPC_E_UNSUPPORTED  -5   /* horizon larger than the kernels cover   */

/* ---- per-instance status bits (ismpc_tick_out.status) -------------------
 * The reference ignores every solver status (utils.cpp:128,363,491); these
 * bits exist so that a batch caller can tell which instances left the regime
 * in which the exact solve and the reference's QP solve coincide.           */
#define ISMPC_ST_OK            0
#define ISMPC_ST_X_INFEASIBLE  1   /* horizontal x QP infeasible (qpOASES rv 37)            */
#define ISMPC_ST_Y_INFEASIBLE  2   /* horizontal y QP infeasible                            */
#define ISMPC_ST_Z_INEQ_ACTIVE 4   /* 0 <= S u <= 1e4 (MPCSolver.cpp:158-160) was active:
                                      solved by the dual active-set fallback               */
#define ISMPC_ST_BAD_INDEX     8   /* midpoint window [idx, idx+2N) outside the plan
                                      (MPCSolver.cpp:259,381 would read out of range):
                                      state passed through unchanged                        */
#define ISMPC_ST_FLIGHT        16  /* lambda_0 <= gate: Stage 3 skipped, u_x = u_y = 0
                                      (MPCSolver.cpp:322,402-403) -- not an error           */
#define ISMPC_ST_TICK_SKIPPED  32  /* controlIter % (int)(100 dt) != 0 (MPCSolver.cpp:214):
                                      state passed through -- not an error                  */
#define ISMPC_ST_Z_NAN         64  /* NaN guard fired (MPCSolver.cpp:277-278)               */
#define ISMPC_ST_Z_FAILED      128 /* vertical fallback: iteration limit hit or infeasible (the working set may hold
                                      every row of the horizon); never fed back in closed loops */

/* ---- parameters: parameters.cpp:9-45 and MPCSolver.cpp:253-255 ---------- */
typedef struct ismpc_params {
    int32_t N;                 /* horizon samples           parameters.cpp:42 (=100)       */
    int32_t S;                 /* single-support samples    parameters.cpp:43 (=35)        */
    int32_t F;                 /* double-support samples    parameters.cpp:44 (=10)        */
    int32_t M;                 /* footsteps in horizon      parameters.cpp:45 (unused by solve) */
    double  mpc_dt;            /* mpcTimeStep               parameters.cpp:9               */
    double  control_dt;        /* controlTimeStep           parameters.cpp:10              */
    double  mass;              /* mass_hrp4                 parameters.cpp:39              */
    double  g;                 /* g                         parameters.cpp:40              */
    double  h_des;             /* comTargetHeight           parameters.cpp:16              */
    double  foot_width;        /* footConstraintSquareWidth parameters.cpp:21              */
    double  first_step_halfwidth; /* +-1 m box while footstepCounter<=1, MPCSolver.cpp:334-337 */
    double  q_p, q_u, q_v;     /* vertical QP weights       MPCSolver.cpp:253-255          */
    double  z_ineq_lo, z_ineq_hi; /* bounds on S_bar_z u    MPCSolver.cpp:159-160          */
    double  lambda_gate;       /* 2.0                       MPCSolver.cpp:322,353,406      */
} ismpc_params;

/* Fills *p with the reference's shipped constants (N=100,S=35,F=10,...). */
void ismpc_params_default(ismpc_params* p);

/* ---- per-instance records ----------------------------------------------
 * AoS on purpose: one wavefront owns one instance, so the 72-byte input
 * record is one coalesced read and the 80-byte output record one write.    */
typedef struct ismpc_tick_in {
    double  com_pos[3];        /* State::comPos   types.hpp:8  (x,y,z)                      */
    double  com_vel[3];        /* State::comVel   types.hpp:9                               */
    double  simulation_time;   /* WalkState::simulationTime  types.hpp:79 (frames)          */
    int32_t mpc_iter;          /* WalkState::mpcIter         types.hpp:80                   */
    int32_t control_iter;      /* WalkState::controlIter                                    */
    int32_t footstep_counter;  /* WalkState::footstepCounter                                */
    int32_t reserved;          /* parameter set of the instance (ismpc_create_sweep); 0 otherwise */
} ismpc_tick_in;               /* 72 bytes */

typedef struct ismpc_tick_out {
    double  com_pos[3];        /* next.comPos  MPCSolver.cpp:275,419,421                    */
    double  com_vel[3];        /* next.comVel  MPCSolver.cpp:276,420,422                    */
    double  u0[3];             /* first decision variable of the z, x, y QP
                                  (MPCSolver.cpp:274,402,403): force, ZMP x, ZMP y          */
    int32_t status;            /* ISMPC_ST_* bits                                           */
    int32_t iters;             /* x iterations | y iterations << 8 | z fallback its << 16   */
} ismpc_tick_out;              /* 80 bytes */

typedef struct ismpc_handle ismpc_handle;

/* MPCSolver::MPCSolver (MPCSolver.cpp:5-200).  `ftsp` is the caller's
 * ftsp_and_timings matrix, rows x 4 row-major (x, y, z, t), as built at
 * Controller.cpp:89-97.  The plan is captured here exactly as the reference
 * captures it at construction (solve() never re-reads it, MPCSolver.cpp:441).
 * `device` is the HIP device ordinal.  Does not read the ../vertical_motion z.txt and f.txt files
 * (MPCSolver.cpp:8-29 loads them but solve() never uses the values).        */
int ismpc_create(const ismpc_params* params, const double* ftsp, int rows,
                 int device, ismpc_handle** out);

/* PARAMETER SWEEPS.  BASELINE's data-parallel axis for this path is "parameter sweeps / Monte-Carlo perturbations of
 * the same horizon": `n_sets` parameter sets that share N, S, F, M, mpc_dt, control_dt, g and lambda_gate and may differ
 * in mass, q_p / q_u / q_v, h_des, foot_width, first_step_halfwidth and the bounds on S u (the reference re-compiles
 * parameters.cpp:9-45 / MPCSolver.cpp:253-255 and re-runs per value).  Every set needs its own inverse of the vertical
 * Hessian MPCSolver.cpp:258 and the tables derived from it; for a sweep they are built ON THE DEVICE for all sets at
 * once -- Newton-Schulz inverse as batched dense products on v_mfma_f64_16x16x4_f64 (csrc/ismpc_sweep.hip).  An instance
 * names its set in ismpc_tick_in.reserved (0 <= reserved < n_sets; anything else: ISMPC_ST_BAD_INDEX, state passed
 * through).  Any plan and any horizon a plain handle takes (N <= 256: the lane-group kernels up to 128, one instance per
 * wavefront beyond; a plan with footsteps off z = 0 adds per-frame offset tables per set, also built on the device); F <= 16.
 * All entry points below work on a sweep handle.                                                                      */
int ismpc_create_sweep(const ismpc_params* params, int n_sets, const double* ftsp, int rows,
                       int device, ismpc_handle** out);
/* Optional, for batches whose instance -> set assignment stays put from call to call (a sweep's usual shape): sorts the instances
 * of the `batch` records at in_dev by their parameter set, once (a counting sort on the device; returns when done).  Later
 * ismpc_solve_batch* calls of the SAME batch size then run instance order[g] in launch slot g, so that the lane groups of a
 * wavefront read one set's tables, and hand each XCD (workgroup index mod 8) one contiguous eighth of the sorted batch: K / 8
 * sets' tables per L2 instead of all K.  Only the placement of instances in the launch changes: every record is byte-identical
 * with and without it, also when `reserved` has changed since (each instance still reads its own set; a stale order only costs
 * the locality).  Rebind after changing the assignment or the batch size; batch = 0 unbinds.                                     */
int ismpc_sweep_bind(ismpc_handle* h, int batch, const ismpc_tick_in* in_dev, void* stream);
/* n_sets (1 for a plain handle), Newton-Schulz iterations run, batched MFMA product launches, table build time. */
int ismpc_sweep_info(const ismpc_handle* h, int* n_sets, int* newton_iterations, int* mfma_gemm_launches, double* build_ms);
/* The device-built tables of one set against the host's long-double build of the same parameters (what ismpc_create
 * does for one set): rel_err[t] = max |device - host| / max |host| for t = 0 H^-1, 1 affine tables of the vertical stage,
 * 2 W_p, 3 S W_p, 4 Hinv S', 5 S Hinv S', 6 anticipative tails, 7 the lane-group layout of 1.  rel_err: 8 doubles.   */
int ismpc_sweep_verify_tables(ismpc_handle* h, int set, double* rel_err);

void ismpc_destroy(ismpc_handle* h);

/* One MPCSolver::solve per instance, `batch` independent instances.
 * Host pointers; copies in, runs, copies out, returns when done
 * (Controller.cpp:346-348: by value in, by value out).
 * Records in PAGE-LOCKED buffers (hipHostMalloc, hipHostRegister, or the
 * ismpc_host_* helpers below) are read and written by the kernel IN PLACE
 * over PCIe (zero copy: no staging, no DMA submissions; reads and writes use
 * the two directions of the link at once); results are bit-identical to the
 * device-pointer entry point.  Pageable buffers work too: they are staged
 * through device memory (the HIP runtime pins them chunk by chunk), and
 * batches of <= 64 records go through a small mapped staging block.        */
int ismpc_solve_batch(ismpc_handle* h, int batch,
                      const ismpc_tick_in* in_host, ismpc_tick_out* out_host);

/* Page-locked host memory without HIP headers on the caller's side: allocate
 * record buffers with ismpc_host_alloc, or pin existing ones for as long as
 * they live with ismpc_host_register (unregister BEFORE freeing them).  The
 * kernel's writes are visible to the host when ismpc_solve_batch returns.    */
int ismpc_host_alloc(size_t bytes, void** out);
int ismpc_host_free(void* p);
int ismpc_host_register(void* p, size_t bytes);
int ismpc_host_unregister(void* p);

/* Same, device pointers, enqueued on `stream` (a hipStream_t; NULL = the
 * default stream), asynchronous.  `u_traj` is NULL or a device buffer of
 * batch x 3 x N doubles receiving the three decision trajectories
 * (decisionVariables_z/_x/_y, MPCSolver.cpp:269,395,396).
 * A handle owns scratch that its launches share (the per-instance marks of the
 * inequality fallback): launches of ONE handle must be ordered -- one stream, or
 * streams synchronised by the caller.  Independent handles are independent.
 * When that scratch has to grow inside a call whose stream is not the previous
 * call's, the previous stream is drained first (ismpc_reserve avoids both).
 * Reproducibility: an instance's record does not depend on where it sits in the
 * batch nor on the other instances, byte for byte, as long as the batch stays in
 * one size class (<= 2 048, <= 8 192, larger: the kernels use 32, 16 and 8 lanes
 * per instance, which sum in different orders); across classes records agree to
 * rounding (~1e-15 relative).  ISMPC_LPI=8|16|32 fixes one layout for all sizes. */
int ismpc_solve_batch_device(ismpc_handle* h, int batch,
                             const ismpc_tick_in* in_dev, ismpc_tick_out* out_dev,
                             double* u_traj, void* stream);

/* Closed loop on the device: for t = 0..ticks-1 run the caller bookkeeping
 * of Controller.cpp:297-304 (enabled, i.e. without the `&& false`), set
 * simulationTime = first_frame + t (Controller.cpp:310), solve, feed the
 * output back as the next input (Controller.cpp:346-348), then
 * ++controlIter, mpcIter = floor(controlIter*cdt/dt) (Controller.cpp:503-504).
 * `state` (device, batch records) is updated in place; `traj` is NULL or a
 * device buffer of ticks x batch ismpc_tick_out records.
 * The kernels pick the number of lanes per instance by batch size, and the
 * closed loop and the single tick pick differently beyond 8 192 instances per
 * call: a rollout tick and an ismpc_solve_batch* tick of the same state then
 * agree to rounding (sums taken in another order), not to the byte.          */
int ismpc_rollout_device(ismpc_handle* h, int batch, ismpc_tick_in* state_dev,
                         int first_frame, int ticks, ismpc_tick_out* traj_dev,
                         void* stream);

/* Optional: size the handle's per-launch scratch for batches up to max_batch now (synchronous).  Without it the
 * scratch grows inside the asynchronous entry points with stream-ordered allocations (hipMallocAsync on the caller's
 * stream: no device-wide synchronisation); callers that capture the launches into a hipGraph call this first.  */
int ismpc_reserve(ismpc_handle* h, int max_batch);

/* Introspection. */
int         ismpc_abi_version(void);
const char* ismpc_last_error(void);        /* thread-local, never NULL */
int         ismpc_get_params(const ismpc_handle* h, ismpc_params* out);
int         ismpc_midpoint_rows(const ismpc_handle* h);   /* rows*(S+F), MPCSolver.cpp:167 */
/* Copies the host copy of ftsp_midpoint (rows*(S+F) x 3 row-major) into dst. */
int         ismpc_get_midpoint(const ismpc_handle* h, double* dst, int capacity_rows);
/* Last kernel time of ismpc_solve_batch* in milliseconds measured with HIP
 * events on the launch stream (0 when timing is disabled).                 */
/* The handle's four self-resetting counters of the inequality fallback, after draining the device: entries in the
 * deferred list, fallback workgroups done, instances parked by an in-kernel rollout, resume workgroups done.  All four
 * are 0 between calls, whatever was launched before (ticks, rollouts, hipGraph replays of a captured step).          */
int         ismpc_fallback_counters(ismpc_handle* h, int* out4);
int         ismpc_set_timing(ismpc_handle* h, int enabled);
double      ismpc_last_kernel_ms(ismpc_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* ISMPC_H */
