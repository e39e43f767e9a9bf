// Shared by the Formulation A translation units: the per-handle device constants and the launch record the host side
// (ismpc_a_hip.hip) hands to the wavefront-per-QP kernels (ismpc_a_wave.hpp, one translation unit per rows-per-lane value so
// that hipcc compiles them side by side).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ismpc_a.h"

namespace ismpc_a {

constexpr int WG = 256;                // threads per workgroup; requires C + F <= 256
constexpr int MAXF = 8;

struct DevA {
    int C, P, F, step, ds, n_gait, ncl, ldq, max_iter, sinv_in_lds;
    int warm_add, warm_drop, warm_extra, warm_min_viol; // block warm start of the wave kernel: passes that add + drop rows, passes that only drop,
                                         // re-entries into adding passes, violated rows that make a re-entry worth it
    int warm_gi, warm_peel_end;          // Goldfarb-Idnani row additions BEFORE the block passes (0: passes start from the equality-only
                                         // point); 1: a negative run end takes more than itself with it (0: only itself)
    int warm_rounds, warm_round_adds;    // after the passes: up to warm_rounds times, warm_round_adds more Goldfarb-Idnani additions and, if
                                         // rows are still violated then, the passes again (a collapsed warm start recovers in one round)
    double dt, eta, w, Qf, disp_forw, disp_forw_dummy, disp_L, aa, wP, sumw;
    double Au[9], Bu[3];
    const double *a, *PA, *PA2, *wtail; // stability row, prefix sums PA[i] = sum_{k<i} a_k and PA2[i] = sum_{k<i} a_k^2, tail weights (index i-(C+1))
    const double *fsx, *fsy;           // base plan, 0-based (fs_plan(k+1))
    const double *clx0, *cly0, *clx1, *cly1;   // centreline: initial / rebuilt structure, 0-based (cl(k+1))
    double* scratch;                   // per-workgroup S^-1 : ldq x ldq doubles each
    const double *plan_x[4], *plan_y[4]; int nplans; double grav;   // base plans selectable per instance (ismpc_a_inst.plan)
    // anticipative tail of the handle's own gait parameters as a function of the tick index alone (quad_walk_no_plots.m:227-231):
    // T[j] = sum_{i=C+1..P} wtail_i cl(j+i) + wP cl(P), per centreline table; the tail of a tick is T[j] + (offset - current footstep) sumw
    const double *tlx0, *tly0, *tlx1, *tly1;
    // reciprocals and roots of the handle's own parameters the wave kernel would otherwise form once per QP on all 64 lanes
    double sqQf, isqQf, iQf, ieta, inv_ds; float rstep;
};

// Per-instance gait parameters (ismpc_a_inst): what depends on the instance alone and costs transcendental functions, divisions or
// reductions -- eta, lambda = exp(-eta dt) and its powers, the coefficients of the stability row a_i = k1c lambda^i - k2c
// (quad_walk_no_plots.m:233-238) and of its prefix sums in closed form
//     sum_{k<i} a_k   = A1 (1 - lambda^i) - i k2c,      sum_{k<i} a_k^2 = A2 (1 - lambda^2i) - B2 (1 - lambda^i) + i k2c^2
// -- computed by one thread per instance in the tick prologue instead of by all 64 lanes of each of its two QPs (round 4).
struct PiPre {
    double eta, lam, lamC, lamP, k1c, k2c, A1, A2, B2, aa;
    double sqQf, isqQf, iQf, ieta, inv_ds, inv_dsm1;     // sqrt(Qf), 1 / sqrt(Qf), 1 / Qf, 1 / eta, 1 / ds, 1 / (ds - 1)
    double ch, sh, sh_eta;                               // cosh(eta dt), sinh(eta dt), sinh(eta dt) / eta: the LIP update (:67-71)
    float rstep; int pad_;                               // 1 / step
};

// One launch of the wavefront-per-QP kernel.  precision: 0 = the QP is solved in fp64, 1 = in fp32 (state, right-hand sides
// and the LIP update stay fp64).
struct WaveLaunch {
    const DevA* c_dev; int F;                      // the handle's constants in device memory; footsteps in the horizon (kernel shape)
    const ismpc_a_state* prev; ismpc_a_state* state; const ismpc_a_inst* inst; const double* push; ismpc_a_out* out;
    int batch; int* work_counter; unsigned long long* hist; int hist_load;
    int precision; int cus; int* occ_cache;       // occ_cache: per handle, [F - 3][precision x per-instance], 0 = not queried yet
    const int* order; const int* count_ptr;       // NULL, or the instances of this launch (device list + device count): see tick_launch
    int claim_chunk;                              // QPs a wavefront takes from the work counter per atomic
    int static_q;                                 // sixteenths of the launch's QPs that are dealt out statically (no atomics) first
    int order_is_qp;                              // order[] lists QPs (2 instance + axis) instead of instances: the fp64 re-solve of the few QPs
    int* defer_list; int* defer_count;            // ... whose fp32 block solve failed its check (NULL: such a QP starts cold instead)
    int grid_cap;                                 // > 0: at most this many workgroups (the re-solve launch)
    hipStream_t stream;
    const PiPre* pre = nullptr;                   // per-instance launches (inst != NULL): the prologue's record of every instance
};

// Implemented in ismpc_a_wave_rl{2,3,4}.hip.  Return 0, -1 (no instantiation for this F) or -2 (HIP error, *err set).
int launch_wave_rl2(const WaveLaunch& L, hipError_t* err);
int launch_wave_rl3(const WaveLaunch& L, hipError_t* err);
int launch_wave_rl4(const WaveLaunch& L, hipError_t* err);

}  // namespace ismpc_a
