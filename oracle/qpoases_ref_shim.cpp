// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// extern "C" door into the reference's own vendored qpOASES 3.2
// (/root/reference/AMR_code_DART/qpOASES/*.cpp, compiled in place by
// oracle/Makefile into oracle/_ref/libqpoases_ref.so; the sources are never
// copied into this repository).
//
// The one entry point reproduces the reference's own calling convention for
// this solver, AMR_code_DART/utils.cpp:121-130 (`solveQP`): Options::setToMPC,
// printLevel = PL_NONE, nWSR = 300, a fresh QProblem(nV, nC) per call, cold
// init(H, g, A, 0, 0, lbA, ubA, nWSR, ...), getPrimalSolution.  H and A are
// dense row-major (utils.cpp:104-117); there are no variable bounds.
#include <qpOASES.hpp>

extern "C" int qpoases_ref_solve(int nV, int nC, const double* H, const double* g,
                                 const double* A, const double* lbA, const double* ubA,
                                 double* x, int* nWSR_inout)
{
    qpOASES::Options options;
    options.setToMPC();
    options.printLevel = qpOASES::PL_NONE;
    qpOASES::int_t nWSR = (nWSR_inout && *nWSR_inout > 0) ? *nWSR_inout : 300;

    qpOASES::QProblem qp(nV, nC);
    qp.setOptions(options);
    qpOASES::returnValue rv =
        qp.init(H, g, A, 0, 0, lbA, ubA, nWSR, NULL, NULL, NULL, NULL, NULL, NULL);
    qp.getPrimalSolution(x);
    if (nWSR_inout) *nWSR_inout = (int)nWSR;
    return (int)rv;
}

extern "C" const char* qpoases_ref_version(void) { return "qpOASES 3.2 (reference vendored copy)"; }
