// Host-side, one-off precompute of everything MPCSolver::MPCSolver
// (reference AMR_code_DART/MPCSolver.cpp:5-200) and the constant part of
// MPCSolver::solve (:220-264) produce, re-derived for the batched HIP path:
// nothing here is an N x N Toeplitz matrix any more, because the kernels apply
// S_bar_z / S_bar_z_v as prefix sums.  What stays dense is the inverse of the
// (constant) vertical Hessian and the per-mpcIter equality corrections.
#pragma once
#include <vector>
#include <string>
#include "../../include/ismpc.h"

namespace ismpc {

struct Tables {
    ismpc_params p{};
    double eta = 0;          // sqrt(g / h_des), parameters.cpp:41
    int rows = 0;            // footstep rows
    int nmid = 0;            // rows*(S+F), MPCSolver.cpp:167
    int NP = 0;              // N rounded up to 16 (MFMA tile)
    int npat = 0;            // S+F equality patterns (one per mpcIter), MPCSolver.cpp:223-243
    int Fmax = 0;            // max equality rows of a pattern (= F)
    int tick_divisor = 1;    // (int)(100*mpcTimeStep), MPCSolver.cpp:214
    std::vector<double> Hinv;     // NP x NP row-major, zero padded: (q_p S'S + q_v Sv'Sv + q_u I)^-1, MPCSolver.cpp:258
    std::vector<double> W;        // npat x Fmax x NP: Hinv[:,E] (Hinv[E,E])^-1, column e of pattern p at W[(p*Fmax+e)*NP + n]
    std::vector<int>    e_lo, ne; // npat each: equality index range [e_lo, e_lo+ne) clipped to the horizon
    std::vector<double> midx, midy, midz;   // nmid each: ftsp_midpoint columns, MPCSolver.cpp:167-180
    std::vector<double> tailx, taily;       // nmid each: eta*dt*sum_i exp(-dt*eta*i)*mid[idx+N+i], MPCSolver.cpp:183-184,381-383
    std::vector<double> ftsp_t;             // rows: ftsp_and_timings(:,3), Controller.cpp:96

    // ---- "affine" form of the vertical stage (what the fast kernel reads) ----
    // With the inequality rows inactive, the z-QP (MPCSolver.cpp:258-269) is an equality constrained
    // least squares whose solution is AFFINE in the state (z, zdot) and in the mid_z window:
    //     u      = U0_p  + z Ua_p  + zdot Ub_p   [+ dU(idx)  - W_p  (dU(idx))_E]
    //     S_z u  = SU0_p + z SUa_p + zdot SUb_p  [+ SdU(idx) - SW_p (dU(idx))_E]
    // p = equality pattern (mpcIter, or npat = "no equalities": footstepCounter <= 1 or mpcIter >= S+F).
    // The dense N x N contraction with Hinv therefore happens here, once, not per tick.
    static constexpr int NT = 256;          // padded horizon of every table row
    std::vector<double> vtab;               // (npat+1) x 6 x NT : U0,Ua,Ub,SU0,SUa,SUb ; zero for n >= N
    std::vector<double> tz, tg;             // NT each: T_bar_z(n,1) = (n+1) dt ; T_bar_g_z(n) = -g dt^2 n(n+1)/2
    bool flat = true;                       // ftsp_midpoint(:,2) == 0 everywhere (the reference's plan, Controller.cpp:95)
    std::vector<double> dU, SdU;            // nmid x NT each (only when !flat): Hinv q_p S' mid_z[idx:idx+N] and S_z times it
    std::vector<double> SW;                 // npat x Fmax x NT: S_z W_p
    std::vector<double> Wt;                 // npat x Fmax x NT: W_p re-strided to NT
    // inequality fallback (0 <= S u <= 1e4 active): row k holds Hinv S_k' / S Hinv S_k'
    std::vector<double> HSt, SHSt;          // N x NT each
};

// Returns ISMPC_OK or an ISMPC_E_* code; on error `err` explains.
int build_tables(const ismpc_params& p, const double* ftsp, int rows, Tables& out, std::string& err);

}  // namespace ismpc
