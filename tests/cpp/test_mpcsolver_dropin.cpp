// GPU test program (run by tests/test_gpu_dropin.py): the C++ MPCSolver drop-in, used exactly the way
// the reference's Controller uses its MPCSolver -- plan of Controller.cpp:89-97, `new MPCSolver(ref)`
// (:105-106), then `desired = solver->solve(desired, walkState, ref)` (:346-348) with the caller
// bookkeeping of :297-304 (enabled), :310, :503-504 -- printing one line per frame for the checker.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "MPCSolver.hpp"

int main(int argc, char** argv)
{
    const int horizon = argc > 1 ? std::atoi(argv[1]) : 100;
    const int frames = argc > 2 ? std::atoi(argv[2]) : 120;
    ismpc_params p; ismpc_params_default(&p); p.N = horizon;
    const int rows = 40;
    Eigen::MatrixXd ftsp = Eigen::MatrixXd::Zero(rows, 4);               // Controller.cpp:89-97
    for (int i = 1; i < rows; ++i) {
        ftsp(i, 0) = (i - 1) * 0.2;
        ftsp(i, 1) = std::pow(-1.0, (double)(i - 1)) * 0.08;
        ftsp(i, 2) = 0.0;
        ftsp(i, 3) = (p.mpc_dt / p.control_dt) * (p.S + p.F) * i;
    }
    MPCSolver* solver = new MPCSolver(ftsp, &p, 0);                       // Controller.cpp:105-106
    State desired; desired.comPos = Eigen::Vector3d(0.0, 0.0, p.h_des);   // Controller.cpp:110
    desired.leftBackFootPos = Eigen::Vector3d(1.0, 2.0, 3.0);             // must pass through untouched
    WalkState ws; ws.mpcIter = 0; ws.controlIter = 0; ws.footstepCounter = 0; ws.supportFoot = true;   // :65-68
    ws.simulationTime = 0; ws.indInitial = 0;
    for (int frame = 0; frame < frames; ++frame) {
        if (ws.simulationTime >= ftsp(ws.footstepCounter, 3) - 1) {       // :297-304 without `&& false`
            ws.controlIter = 0; ws.mpcIter = 0; ws.footstepCounter++; ws.supportFoot = !ws.supportFoot;
        }
        ws.simulationTime = frame;                                        // :310
        desired = solver->solve(desired, ws, ftsp);                       // :346-348
        std::printf("%d %d %d %.17g %.17g %.17g %.17g %.17g %.17g %d\n", frame, ws.footstepCounter, ws.mpcIter,
                    desired.comPos(0), desired.comPos(1), desired.comPos(2),
                    desired.comVel(0), desired.comVel(1), desired.comVel(2), solver->last_status);
        if (solver->itr != ws.mpcIter || solver->fsCount != ws.footstepCounter) return 3;
        ++ws.controlIter;                                                 // :503
        ws.mpcIter = (int)std::floor(ws.controlIter * p.control_dt / p.mpc_dt);   // :504
    }
    if (desired.leftBackFootPos(0) != 1.0 || desired.leftBackFootPos(2) != 3.0) return 4;
    delete solver;
    return 0;
}
