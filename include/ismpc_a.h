/*
 * ismpc_a.h -- C ABI of the MI355X-native "Formulation A" ISMPC tick (classic ISMPC with automatic
 * footstep adaptation): the per-tick QP of the reference's MATLAB gait generators, which produced every
 * checked-in trajectory fixture and whose stacked matrices the C++ MPCSolver constructor still allocates
 * (AMR_code_DART/MPCSolver.cpp:34-71).  Reference interfaces replaced (paths relative to the reference root):
 *
 *   ismpc_a_plan            <->  trotting/init_quadruped.m:5-184 , walking/init_quadruped2.m:5-284
 *                                (foot_plan, center = fs_plan)
 *   ismpc_a_create          <->  walking/quad_walk_no_plots.m:6-110 / trotting/quad_as_bip_no_plots.m:6-103
 *                                (constants, A_upd/B_upd, centreline cl_x/cl_y)
 *   ismpc_a_tick_batch_device <-> one iteration of `for j = 1:sim_duration`:
 *                                walking/quad_walk_no_plots.m:127-331,509-559 /
 *                                trotting/quad_as_bip_no_plots.m:116-316,436-479
 *                                (mapping, ZMP / kinematic / stability constraints, quadprog, LIP update,
 *                                 footstep counter + plan shift + centreline rebuild)
 *   ismpc_a_rollout_device  <->  the whole loop, closed on the device
 *
 * The swing-foot re-placement QPs, the foot trajectories and the text wire format are the
 * ismpc_a_*feet* / ismpc_a_foot_trajectories / ismpc_a_write_trajectory_txt entry points further down.
 * Conventions as ismpc.h: plain C, int status, no CPU fallback; every entry point runs on the handle's device and restores
 * the caller's current one.  Launches of one handle must be ordered (the handle owns the copy of the previous state, the
 * work counter and the working-set history that its launches share): use one stream per handle, or order the streams
 * yourself.  When that scratch has to grow inside an asynchronous entry point and the call's stream is not the previous
 * call's, the previous stream is drained first (ismpc_a_reserve avoids both the growth and the wait).
 */
#ifndef ISMPC_A_H
#define ISMPC_A_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-instance status bits (ismpc_a_out.status) */
#define ISMPC_A_ST_OK            0
#define ISMPC_A_ST_X_INFEASIBLE  1    /* the x QP has no feasible point                         */
#define ISMPC_A_ST_Y_INFEASIBLE  2
#define ISMPC_A_ST_OVERFLOW      4    /* the horizon spans more than F-1 step boundaries: the .m file's
                                         `mapping` outgrows its F+1 columns (needs F = ceil(C/step)+1) */
#define ISMPC_A_ST_BAD_INDEX     8    /* j + P beyond the centreline, fc + F beyond the plan, or j outside
                                         the current step [step (fc-1), step fc - 1]               */
#define ISMPC_A_ST_ITER_LIMIT    16   /* active-set iteration limit hit (result is the last iterate)  */
#define ISMPC_A_ST_UNVERIFIED    32   /* set together with X/Y_INFEASIBLE when the flag comes from the final check of
                                         the returned point (a row, a kinematic limit or the stability row is off by
                                         more than 1e-7 relative) and not from the active-set logic itself */

typedef struct ismpc_a_gait {           /* init_quadruped*.m:5-37 */
    int32_t gait;                       /* 0 = trot (init_quadruped.m), 1 = walk (init_quadruped2.m) */
    int32_t n_gait;                     /* N_gait = 100 */
    double  disp_A, phi;                /* step length [m], heading [rad] */
    double  disp_B, disp_C;             /* 0.259394, 0.88 */
    double  disp_i, disp_o, disp_forw;  /* 0.4, 0.4, 0.5 */
} ismpc_a_gait;

typedef struct ismpc_a_params {         /* quad_walk_no_plots.m:15-45,270-271 */
    int32_t C, P, F;                    /* control / preview horizon, footsteps: 100/200/3 (walk), 160/320/3 (trot) */
    int32_t step, ds;                   /* step_duration, dsSamples: 50/30 (walk), 80/50 (trot) */
    int32_t n_gait;                     /* NF */
    double  dt;                         /* mpcTimeStep */
    double  height;                     /* 0.56 */
    double  grav;                       /* 9.8 (the scripts do not use 9.81) */
    double  w;                          /* centroid_size = foot_size = 0.02 */
    double  Qf;                         /* Qfootsteps: 1e9 (walk), 1e7 (trot) */
    double  disp_forw, disp_forw_dummy, disp_L;   /* 0.5, 0.25, (disp_o + disp_i)/2 */
} ismpc_a_params;

/* everything the MATLAB loop carries from tick to tick, per instance (96 bytes) */
typedef struct ismpc_a_state {
    double  x, xd, xz, y, yd, yz;       /* CoM, CoM velocity, ZMP */
    double  cur_x, cur_y;               /* current_xfs, current_yfs */
    double  off_x, off_y;               /* fs_plan(:,c) = base plan + off (quad_walk_no_plots.m:535-536) */
    int32_t fc;                         /* fsCounter, 1-based like the script */
    int32_t j;                          /* tick about to run, 1-based */
    int32_t rebuilt;                    /* 0: initial centreline (:86-99), 1: rebuilt one (:540-549) */
    int32_t reserved;
} ismpc_a_state;

typedef struct ismpc_a_out {            /* 80 bytes */
    double  com_before[2];              /* x_store(j), y_store(j): row j of ComTrajectory_*.txt */
    double  vel_after[2];               /* xd_store(j), yd_store(j): row j of ComVelocity_*.txt */
    double  u0[2];                      /* predicted_xzd(1), predicted_yzd(1) */
    double  f0[2];                      /* predicted_xfs(1), predicted_yfs(1) */
    int32_t status;
    int32_t iters_x, iters_y;           /* work per QP: block warm-start passes + Goldfarb-Idnani steps (+1 for a polish solve) */
    int32_t active;                     /* final working-set sizes: x | y << 16 */
} ismpc_a_out;

/* Per-instance gait parameters for Monte-Carlo / parameter-sweep batches (BASELINE.json configs[4]): every field
 * overrides the handle-wide value of ismpc_a_params for one instance (32 bytes).  The .m scripts hard-code these at
 * quad_walk_no_plots.m:20-45 / quad_as_bip_no_plots.m:16-39; a sweep re-runs the script once per value. */
typedef struct ismpc_a_inst {
    double  height;                     /* CoM height: eta = sqrt(grav / height), A_upd / B_upd, the stability row */
    double  Qf;                         /* Qfootsteps */
    int32_t step, ds;                   /* step_duration, dsSamples */
    int32_t F;                          /* footsteps in this instance's horizon, 1 <= F <= ismpc_a_params.F of the handle */
    int32_t plan;                       /* base plan: 0 = the one given to ismpc_a_create, k = k-th ismpc_a_add_plan */
} ismpc_a_inst;

typedef struct ismpc_a_handle ismpc_a_handle;

void ismpc_a_params_default(int gait, ismpc_a_params* p);
void ismpc_a_gait_default(int gait, double phi, double disp_A, ismpc_a_gait* g);

/* Host: the plan generators.  foot_plan: (n_gait+1) x 8 row-major (BL, BR, FR, FL xy); center: n_gait x 2.
 * Returns the number of foot_plan rows written (n_gait for trot, n_gait+1 for walk) or a negative error. */
int ismpc_a_plan(const ismpc_a_gait* g, double* foot_plan, double* center);

/* center = fs_plan (n_gait x 2 row-major).  disp_C sets the initial state (x = xz = disp_C/2). */
int ismpc_a_create(const ismpc_a_params* p, const double* center, int device, ismpc_a_handle** out);
void ismpc_a_destroy(ismpc_a_handle* h);

/* Optional: size the handle's scratch (copy of the previous state, working-set history) for batches up to max_batch now;
 * otherwise it grows inside the asynchronous entry points with stream-ordered allocations on the caller's stream. */
int ismpc_a_reserve(ismpc_a_handle* h, int max_batch);

/* State the scripts start from (quad_walk_no_plots.m:52-62). */
int ismpc_a_initial_state(const ismpc_a_handle* h, double disp_C, ismpc_a_state* st);

/* One tick for every instance.  state is updated in place; push is NULL or batch x 2 impulsive velocity
 * disturbances added before the QP (quad_walk_no_plots.m:134-148); out may be NULL. */
int ismpc_a_tick_batch_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const double* push_dev,
                              ismpc_a_out* out_dev, void* stream);
/* `ticks` ticks, out_traj NULL or ticks x batch records.  Inside a rollout every QP starts from the working set the same
 * instance ended the previous tick with (moved by one sample); the optimum is the same, the route shorter. */
int ismpc_a_rollout_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, int ticks,
                           ismpc_a_out* out_traj_dev, void* stream);
/* Arithmetic type of the QP solve: fp32 = 0 (default) solves in fp64, fp32 = 1 in fp32 -- the dtype BASELINE.json names
 * for its walking-gait and Monte-Carlo configurations.  In both cases the QP is posed relative to the current footstep
 * (every number the solver touches is step-sized), its right-hand sides are formed in fp64 from the fp64 state, and the
 * LIP update of the state (quad_walk_no_plots.m:297-322) is fp64; with fp32 = 1 the active-set iterations, the small
 * linear systems and the returned u0 / f0 carry fp32 rounding (relative CoM error of a tick stays below 1e-6: a tick moves
 * the CoM by B_upd u0 with |B_upd| ~ 1e-6..1e-3).  A QP whose working set pins (nearly) the whole horizon is beyond the
 * fp32 block solve; it is solved by the fp64 instantiation in a small launch (up to 64 workgroups) that follows every fp32 launch on the
 * same stream (about one QP in 30 000 at pushes 1.5x the bench's).  Needs 3 <= F <= 6. */
int ismpc_a_set_precision(ismpc_a_handle* h, int fp32);

/* Introspection (synchronises the stream of the last launch): QPs the last fp32 launch handed to the fp64 re-solve. */
int ismpc_a_last_deferred(ismpc_a_handle* h);

/* The same first guess for caller-driven loops of ismpc_a_tick_batch*_device: enable it when instance i of one call is
 * instance i of the previous one (a wrong guess costs time, never accuracy).  Off by default. */
int ismpc_a_set_warm_history(ismpc_a_handle* h, int enabled);

/* Per-instance gait parameters (one ismpc_a_inst per instance, device pointer).  The handle fixes C, P, dt, w, the
 * kinematic limits and the maximum F; height, Qf, step, ds, F and the base plan come from inst_dev.  An instance
 * whose record is invalid (ds >= step, F out of range, unknown plan, ...) gets ISMPC_A_ST_BAD_INDEX and is left alone. */
int ismpc_a_add_plan(ismpc_a_handle* h, const double* center);      /* returns the plan index (1..3) or a negative error */
int ismpc_a_tick_batch_inst_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev,
                                   const double* push_dev, ismpc_a_out* out_dev, void* stream);
int ismpc_a_rollout_inst_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev,
                                int ticks, ismpc_a_out* out_traj_dev, void* stream);

/* ---- swing-foot re-placement (the scripts' second quadprog) and the trajectory wire format -------------------
 *   ismpc_a_feet_init_device / ismpc_a_tick_feet_batch_device / ismpc_a_rollout_feet_device
 *        <->  trotting/quad_as_bip_no_plots.m:332-426 + compute_two_feet1.m:1-56
 *             walking/quad_walk_no_plots.m:336-504   + compute_one_feet_walk.m:84-140
 *   ismpc_a_foot_trajectories   <->  quad_as_bip_no_plots.m:482-509 / quad_walk_no_plots.m:562-613
 *   ismpc_a_write_trajectory_txt <-> fprintf(file, '%d %d %d\n', row)   (what Controller.cpp:147-281 reads back)
 * Every instance carries its own foot_plan: feet_dev is batch x ismpc_a_feet_rows(h) x 8 doubles
 * (columns BL, BR, FR, FL xy).                                                                           */
#define ISMPC_A_FEET_PAD 8                           /* feet_dev rows per instance = plan rows + ISMPC_A_FEET_PAD   */
int ismpc_a_feet_rows(const ismpc_a_handle* h);     /* rows per instance in feet_dev, once initialised             */
int ismpc_a_feet_init_device(ismpc_a_handle* h, const ismpc_a_gait* g, const double* foot_plan_host, int rows,
                             int batch, double* feet_dev, void* stream);
/* one tick (as ismpc_a_tick_batch_device) followed by the foot QP of every instance */
int ismpc_a_tick_feet_batch_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const double* push_dev,
                                   ismpc_a_out* out_dev, double* feet_dev, void* stream);
int ismpc_a_rollout_feet_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, int ticks,
                                ismpc_a_out* out_traj_dev, double* feet_dev, void* stream);
/* The same for batches with per-instance gait parameters (ismpc_a_inst; BASELINE configs[4]): instance b follows the foot
 * rules of its base plan inst[b].plan (trot or walk, heading, lateral limits) and starts from that plan's foot_plan.
 * gaits: one record per base plan of the handle, in the order ismpc_a_create / ismpc_a_add_plan registered them;
 * foot_plans_host: nplans x rows x 8 doubles.  The scripts re-run once per parameter value
 * (trotting/quad_as_bip_no_plots.m:332-426 + compute_two_feet1.m:1-56, walking/quad_walk_no_plots.m:336-504 +
 * compute_one_feet_walk.m:84-140); the foot files of instance b are ismpc_a_foot_trajectories on its rows of feet_dev
 * with its own step_duration.                                                                                          */
int ismpc_a_feet_init_inst_device(ismpc_a_handle* h, const ismpc_a_gait* gaits, const double* foot_plans_host, int rows,
                                  int nplans, int batch, const ismpc_a_inst* inst_dev, double* feet_dev, void* stream);
int ismpc_a_tick_feet_batch_inst_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev,
                                        const double* push_dev, ismpc_a_out* out_dev, double* feet_dev, void* stream);
int ismpc_a_rollout_feet_inst_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev,
                                     int ticks, ismpc_a_out* out_traj_dev, double* feet_dev, void* stream);
/* Host: the four foot files of one instance.  foot_plan: rows x 8 (as left by the rollout); dst: 4 x n x 3 in the
 * order fl, fr, rl, rr with n = (sim_duration / step) * step rows; returns n. */
int ismpc_a_foot_trajectories(const ismpc_a_gait* g, int step, const double* foot_plan, int rows, int sim_duration, double* dst);
/* Host: write n rows of 3 values exactly as MATLAB's fprintf(file, '%d %d %d\n', row) does (integers as %d,
 * everything else as %e).  Returns 0 or a negative error. */
int ismpc_a_write_trajectory_txt(const char* path, const double* rows3, int n);

const char* ismpc_a_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ISMPC_A_H */
