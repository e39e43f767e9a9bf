// Parameter sweeps of Formulation B on the device: K parameter sets (mass, vertical-QP weights q_p / q_u / q_v, CoM height,
// foot width, bounds on S u) share one plan and one horizon; every set needs its own inverse of the vertical Hessian
//     H_k = q_p S_k'S_k + q_v Sv_k'Sv_k + q_u I                    (reference AMR_code_DART/MPCSolver.cpp:258, constants :253-255)
// and everything derived from it (the affine tables of ismpc_tables.hpp).  For ONE set the host builds them in long double
// (ismpc_tables.cpp); for a sweep they are built HERE, on the GPU, for all sets at once:
//
//   H_k            closed form of the Toeplitz products (sweep_init)
//   H_k^-1         Newton-Schulz  X <- X (2I - H X)  from X0 = I / bound(|H|_inf): nothing but dense N x N x N products, batched
//                  over the sets on v_mfma_f64_16x16x4_f64 with LDS-staged 64 x 16 / 16 x 64 panels (sweep_gemm).  Quadratic
//                  convergence, self-correcting: the limit is accurate to cond(H) eps, as a Cholesky inverse is.
//   Hinv S', S Hinv S'   the two N x N x N products of the inequality fallback -- the same MFMA kernel
//   per equality pattern: W_p = Hinv[:,E] (Hinv[E,E])^-1, the affine tables u = -(I - W_p E')Hinv f, S u  (sweep_patterns: small,
//                  VALU + LDS), lane-group layouts (sweep_layout), anticipative tails per eta_k (sweep_tail)
//
// ismpc_sweep_verify_tables compares any set with the host's long-double build.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "ismpc_tables.hpp"

namespace ismpc {

struct SweepSlabs {                 // one device slab per table kind, set k at slab + k * stride (doubles)
    int K = 0, NG = 0;              // sets; N rounded up to the 64-wide GEMM tile
    double *H = nullptr, *X0 = nullptr, *X1 = nullptr, *T = nullptr;      // NG x NG each (X0 ends up holding H^-1, symmetrised)
    double *M1 = nullptr;           // Hinv S' (scratch of the fallback tables)
    double *U = nullptr, *Ut = nullptr;   // m S' and m S (common to the sets: unit mass), NG x NG
    double *hvec = nullptr;         // 3 x NG per set: Hinv f0, Hinv fa, Hinv fb
    double *vtab = nullptr, *vqT = nullptr, *Wt = nullptr, *SW = nullptr, *HSt = nullptr, *SHSt = nullptr, *tailx = nullptr, *taily = nullptr;
    double* vqT2 = nullptr; size_t s_vqT2 = 0;   // the second lane-group layout of vtab (lpi2 / R2; null without one)
    double *dU = nullptr, *SdU = nullptr; size_t s_dU = 0;   // plans with mid_z != 0 only: nmid x NT per set (sweep_du)
    size_t s_mat = 0, s_vtab = 0, s_vqT = 0, s_W = 0, s_HS = 0, s_tail = 0;   // strides (doubles)
    double* par = nullptr;          // K x 8: mass, q_p, q_u, q_v, h_des, eta, 1 / bound(|H|_inf), g
    int newton_iters = 0, gemm_launches = 0;
    float build_ms = 0.f;
    double max_residual = 0.0;      // max over the sets of |I - H X| (checked by sweep_build: a set above 1e-3 fails the build)
};

// (a plan with mid_z != 0 adds the per-frame offsets dU, SdU of every set: sweep_du)
// Builds every per-set table on `stream` (synchronises before returning).  `t0` = host tables of set 0 (plan, patterns, structure).
// lpi / R: lane-group layout of vqT (ismpc_hip.hip quad_R); lpi2 / R2: a second layout beside it (vqT2; lpi2 = 0: none).  All device
// memory is appended to `allocs`.
int sweep_build(const ismpc_params* sets, int K, const Tables& t0, const double* midx_dev, const double* midy_dev, const double* midz_dev, const int* e_lo_dev,
                const int* ne_dev, int lpi, int R, int lpi2, int R2, hipStream_t stream, SweepSlabs& out, std::vector<void*>& allocs, std::string& err);

}  // namespace ismpc
