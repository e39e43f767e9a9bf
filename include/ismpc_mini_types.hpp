// Eigen-free stand-ins for the boundary types of the reference's MPCSolver interface, used ONLY when the
// reference's own headers are not on the include path (this repository has no Eigen).  Member names and
// meanings are the reference's: AMR_code_DART/types.hpp:7-28 (State), :77-81 (WalkState).  Inside the
// reference tree define ISMPC_WITH_REFERENCE_TYPES and the real <Eigen/Core> + types.hpp are used instead.
#pragma once
#include <cstddef>
#include <vector>

namespace Eigen {

struct Vector3d {
    double v[3] = {0.0, 0.0, 0.0};
    Vector3d() = default;
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    static Vector3d Zero() { return Vector3d(); }
};

class MatrixXd {                       // dense, column-major like Eigen's default
public:
    MatrixXd() = default;
    MatrixXd(long r, long c) : r_(r), c_(c), d_((std::size_t)(r * c), 0.0) {}
    static MatrixXd Zero(long r, long c) { return MatrixXd(r, c); }
    long rows() const { return r_; }
    long cols() const { return c_; }
    double& operator()(long i, long j) { return d_[(std::size_t)(j * r_ + i)]; }
    double operator()(long i, long j) const { return d_[(std::size_t)(j * r_ + i)]; }
private:
    long r_ = 0, c_ = 0;
    std::vector<double> d_;
};

}  // namespace Eigen

// types.hpp:7-28 -- solve() reads comPos / comVel and passes everything else through (MPCSolver.cpp:210)
struct State {
    Eigen::Vector3d comPos, comVel, comAcc, zmpPos;
    Eigen::Vector3d leftBackFootPos, leftBackFootVel, leftBackFootAcc;
    Eigen::Vector3d rightBackFootPos, rightBackFootVel, rightBackFootAcc;
    Eigen::Vector3d leftFrontFootPos, leftFrontFootVel, leftFrontFootAcc;
    Eigen::Vector3d rightFrontFootPos, rightFrontFootVel, rightFrontFootAcc;
    Eigen::Vector3d torsoOrient, leftBackFootOrient, rightBackFootOrient, leftFrontFootOrient, rightFrontFootOrient;
};

// types.hpp:77-81
struct WalkState {
    bool supportFoot;
    double simulationTime;
    int mpcIter, controlIter, footstepCounter, indInitial;
};
