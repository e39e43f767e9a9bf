// Drop-in for the reference's AMR_code_DART/MPCSolver.hpp:16-28 -- same class name, constructor,
// solve() signature and public diagnostic fields -- implemented as a batch of ONE over the C ABI of
// ismpc.h (HIP on gfx950).  Header-only so that it compiles against whatever Eigen the caller has:
//
//   inside the reference tree :  -DISMPC_WITH_REFERENCE_TYPES   (uses <Eigen/Core> and the reference's types.hpp)
//   stand-alone (this repo)    :  ismpc_mini_types.hpp supplies Eigen::MatrixXd / Vector3d / State / WalkState
//
// Differences a maintainer should know (all documented in INTEGRATION.md):
//   * the horizon and every constant of parameters.cpp are run-time ismpc_params (defaults = the reference's);
//   * ../vertical_motion/{z,f}.txt and ../data/debug.txt are NOT needed (MPCSolver.cpp:3,8-29 only load them);
//   * nothing is printed per tick (MPCSolver.cpp:281-283,310,425-427 print 7 lines);
//   * a failed construction throws std::runtime_error instead of exit(1) (MPCSolver.cpp:9-12).
#pragma once
#ifdef ISMPC_WITH_REFERENCE_TYPES
#include <Eigen/Core>
#include "types.hpp"
#else
#include "ismpc_mini_types.hpp"
#endif
#include <stdexcept>
#include <string>
#include <vector>
#include "ismpc.h"

class MPCSolver {
public:
    // MPCSolver.cpp:5 -- the plan is captured here, as in the reference
    explicit MPCSolver(const Eigen::MatrixXd& ftsp_and_timings) : MPCSolver(ftsp_and_timings, nullptr, 0) {}
    // same, with run-time parameters (nullptr = the reference's constants) and a HIP device ordinal
    MPCSolver(const Eigen::MatrixXd& ftsp_and_timings, const ismpc_params* params, int device)
    {
        ismpc_params p;
        if (params) p = *params; else ismpc_params_default(&p);
        const int rows = (int)ftsp_and_timings.rows();
        if (ftsp_and_timings.cols() != 4) throw std::runtime_error("MPCSolver: ftsp_and_timings must be rows x 4");
        std::vector<double> plan((size_t)rows * 4);
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < 4; ++j) plan[(size_t)i * 4 + j] = ftsp_and_timings(i, j);
        const int rc = ismpc_create(&p, plan.data(), rows, device, &h_);
        if (rc != ISMPC_OK) throw std::runtime_error(std::string("MPCSolver: ") + ismpc_last_error());
        old_fsCount = 0; ct = 0; xz_dot = 0.0; yz_dot = 0.0;      // MPCSolver.cpp:98-102
        itr = 0; fsCount = 0; adaptation_memo = 0; ds_samples = 0;
    }
    ~MPCSolver() { ismpc_destroy(h_); }
    MPCSolver(const MPCSolver&) = delete;
    MPCSolver& operator=(const MPCSolver&) = delete;

    // Compute the next desired state starting from the current state -- MPCSolver.cpp:204.
    // The third argument is accepted and unused, as in the reference (:441 reads it into an unused value).
    State solve(State current, WalkState walkState, const Eigen::MatrixXd& /*ftsp_and_timings*/)
    {
        itr = walkState.mpcIter;                 // MPCSolver.cpp:206
        fsCount = walkState.footstepCounter;     // :207
        State next = current;                    // :210
        ismpc_tick_in in;
        for (int c = 0; c < 3; ++c) { in.com_pos[c] = current.comPos(c); in.com_vel[c] = current.comVel(c); }
        in.simulation_time = walkState.simulationTime;
        in.mpc_iter = walkState.mpcIter; in.control_iter = walkState.controlIter;
        in.footstep_counter = walkState.footstepCounter; in.reserved = 0;
        ismpc_tick_out out;
        const int rc = ismpc_solve_batch(h_, 1, &in, &out);
        if (rc != ISMPC_OK) throw std::runtime_error(std::string("MPCSolver::solve: ") + ismpc_last_error());
        for (int c = 0; c < 3; ++c) { next.comPos(c) = out.com_pos[c]; next.comVel(c) = out.com_vel[c]; }
        last_status = out.status; last_u0[0] = out.u0[0]; last_u0[1] = out.u0[1]; last_u0[2] = out.u0[2];
        return next;                             // :500
    }

    // some stuff (MPCSolver.hpp:24-28)
    int itr;
    int fsCount, old_fsCount, adaptation_memo, ds_samples, ct;
    double xz_dot, yz_dot;

    // extras: what the reference prints per tick (f, zmp x, zmp y) and the ISMPC_ST_* bits of the last solve
    int last_status = 0;
    double last_u0[3] = {0.0, 0.0, 0.0};
    ismpc_handle* handle() { return h_; }

private:
    ismpc_handle* h_ = nullptr;
};
