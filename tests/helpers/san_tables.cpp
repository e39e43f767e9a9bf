// Sanitizer driver for the host-side table builder (csrc/ismpc_tables.cpp): g++ -fsanitize=address,undefined, CPU only.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "ismpc.h"
#include "ismpc_tables.hpp"
static void defaults(ismpc_params* p, int N, int S, int F, double dt)
{
    std::memset(p, 0, sizeof(*p));
    p->mpc_dt = dt; p->control_dt = 0.01; p->N = N; p->S = S; p->F = F; p->M = 2;
    p->mass = 50.0; p->g = 9.81; p->h_des = 0.69; p->foot_width = 0.09; p->first_step_halfwidth = 1.0;
    p->q_p = 1005000.0; p->q_u = 0.01; p->q_v = 100.0; p->z_ineq_lo = 0.0; p->z_ineq_hi = 10000.0; p->lambda_gate = 2.0;
}
int main()
{
    const int cases[][3] = {{50, 35, 10}, {100, 35, 10}, {150, 35, 10}, {200, 35, 10}, {37, 35, 10}, {20, 7, 2}, {128, 35, 10}, {256, 35, 10}};
    for (auto& cs : cases)
        for (int stairs = 0; stairs < 2; ++stairs) {
            ismpc_params p; defaults(&p, cs[0], cs[1], cs[2], cs[0] == 20 ? 0.05 : 0.01);
            const int rows = 40;
            std::vector<double> ftsp((size_t)rows * 4);
            for (int i = 0; i < rows; ++i) { ftsp[4 * i] = 0.2 * i; ftsp[4 * i + 1] = (i % 2 ? -0.08 : 0.08); ftsp[4 * i + 2] = stairs ? 0.01 * ((i / 3) % 4) : 0.0; ftsp[4 * i + 3] = (double)i * (cs[1] + cs[2]); }
            ismpc::Tables t; std::string err;
            const int rc = ismpc::build_tables(p, ftsp.data(), rows, t, err);
            std::printf("N=%d S=%d F=%d stairs=%d rc=%d %s npat=%d nmid=%d\n", cs[0], cs[1], cs[2], stairs, rc, err.c_str(), t.npat, t.nmid);
        }
    return 0;
}
