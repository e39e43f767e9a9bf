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
};

// Returns ISMPC_OK or an ISMPC_E_* code; on error `err` explains.
int build_tables(const ismpc_params& p, const double* ftsp, int rows, Tables& out, std::string& err);

}  // namespace ismpc
