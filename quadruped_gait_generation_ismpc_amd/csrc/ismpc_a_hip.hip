// Formulation A (classic ISMPC with footstep adaptation) on gfx950: kernels + the C ABI of include/ismpc_a.h.
//
// The per-axis QP of one instance (walking/quad_walk_no_plots.m:153-293):
//
//   min 1/2 |u|^2 + Qf/2 |f - p|^2      u = ZMP velocities (C), f = footsteps (F)
//   s.t. a'u = b                          stability (anticipative tail)          (:227-242)
//        lo_i <= dt cumsum(u)_i - M_i f <= hi_i     ZMP band around the mapped footstep  (:153-181)
//        -bl_r <= f_r - f_{r-1} <= bu_r             kinematic                           (:187-222)
//
// The reference hands the stacked dense matrices to quadprog (MATLAB) / qpOASES / HPIPM.  Here the Hessian is
// diagonal and every row has a closed form, so a DUAL ACTIVE-SET method in RANGE-SPACE form never builds a matrix
// over the variables.  Two kernels:
//
//  * ismpc_a_tick_wave<Real, RL, F, PI> (ismpc_a_wave.hpp; default): ONE WAVEFRONT per QP, nothing of working-set size is stored.  The Gram
//    block of the active ZMP rows is dt^2 min(i, k) (a random walk's covariance: tridiagonal inverse, only the gaps
//    between consecutive active rows matter) plus a border of rank <= 2F+1 with closed-form rows; block warm start
//    (primal-dual active-set passes, one structured solve per pass) in front of Goldfarb-Idnani; optional
//    per-instance gait parameters; closed-loop first guess from the previous tick.  DESIGN.md section 2.6.
//  * ismpc_a_tick_kernel (ISMPC_A_KERNEL=block, A/B reference and F > 6): one 256-thread workgroup per QP with an
//    explicit S^-1 = (N' H^-1 N)^-1 of working-set size, rank-1 border / Schur updates, two refinement passes.
//
// Results are the unique minimiser: validated against the oracle's null-space Goldfarb-Idnani and the
// reference's qpOASES (tests/).  No CPU fallback.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <string>
#include <vector>
#include <new>
#include <algorithm>
#include "ismpc_a_dev.hpp"

namespace {

using ismpc_a::DevA;
constexpr int T = ismpc_a::WG;         // threads per workgroup; requires C + F <= 256
constexpr int MAXF = ismpc_a::MAXF;
constexpr int QCAP = 264;              // capacity of the working set (>= C + F + 1)

// ---- wave / block primitives ------------------------------------------------------------------
template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ double dpp64(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan_up(double v)   // inclusive prefix sum over the 64 lanes
{
    v += dpp64<0x111, 0xf, true>(0.0, v);
    v += dpp64<0x112, 0xf, true>(0.0, v);
    v += dpp64<0x114, 0xf, true>(0.0, v);
    v += dpp64<0x118, 0xf, true>(0.0, v);
    v += dpp64<0x142, 0xa, false>(0.0, v);
    v += dpp64<0x143, 0xc, false>(0.0, v);
    return v;
}

struct Shared {
    double u[T], zu[T], imp[T], zlo[T], zhi[T], w1[T], w2[T], a[T], PA[T + 1];
    int k1[T];
    double f[MAXF + 1], zf[MAXF + 1], pref[MAXF + 1], klo[MAXF + 1], khi[MAXF + 1];
    int act_row[QCAP]; double act_sgn[QCAP], mu[QCAP], r[QCAP], dp[QCAP];
    int state[T + MAXF + 1];            // per row (1..C+F): 0 free, +1 lower active, -1 upper active
    double red[T]; int redi[T];
    double wsum[8];
    double zfpart[4][MAXF + 1];
    // scalars
    double b, sviol, sg, gamma, npn, t, t1, t2, mu_p, rowval;
    int q, row, drop, flag, iters, status;
};

// inclusive prefix sum over the workgroup (thread order); every thread calls
__device__ __forceinline__ double block_scan_incl(Shared& s, double v, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const double p = wave_scan_up(v);
    if (lane == 63) s.wsum[wave] = p;
    __syncthreads();
    double add = 0.0;
    for (int wv = 0; wv < wave; ++wv) add += s.wsum[wv];
    __syncthreads();
    return p + add;
}
// inclusive prefix sum plus the workgroup total
__device__ __forceinline__ double block_scan_incl_tot(Shared& s, double v, int tid, double& tot)
{
    const int lane = tid & 63, wave = tid >> 6;
    const double p = wave_scan_up(v);
    if (lane == 63) s.wsum[wave] = p;
    __syncthreads();
    double add = 0.0;
    for (int wv = 0; wv < wave; ++wv) add += s.wsum[wv];
    tot = ((s.wsum[0] + s.wsum[1]) + s.wsum[2]) + s.wsum[3];
    __syncthreads();
    return p + add;
}
// sum over the workgroup, same value (bitwise) in every thread
__device__ __forceinline__ double block_sum(Shared& s, double v, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const double p = wave_scan_up(v);
    if (lane == 63) s.wsum[wave] = p;
    __syncthreads();
    const double tot = ((s.wsum[0] + s.wsum[1]) + s.wsum[2]) + s.wsum[3];
    __syncthreads();
    return tot;
}
// minimum of v with its index (ties: smallest index), broadcast to all threads; v = +inf means "no candidate"
__device__ __forceinline__ void block_argmin(Shared& s, double v, int idx, int tid, double& vmin, int& imin)
{
    s.red[tid] = v; s.redi[tid] = idx;
    __syncthreads();
    if (tid < 16) {
        double bv = s.red[tid * 16]; int bi = s.redi[tid * 16];
        for (int k = 1; k < 16; ++k) {
            const double cv = s.red[tid * 16 + k]; const int ci = s.redi[tid * 16 + k];
            if (cv < bv || (cv == bv && ci < bi)) { bv = cv; bi = ci; }
        }
        s.red[tid * 16] = bv; s.redi[tid * 16] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = s.red[0]; int bi = s.redi[0];
        for (int k = 1; k < 16; ++k) {
            const double cv = s.red[k * 16]; const int ci = s.redi[k * 16];
            if (cv < bv || (cv == bv && ci < bi)) { bv = cv; bi = ci; }
        }
        s.red[0] = bv; s.redi[0] = bi;
    }
    __syncthreads();
    vmin = s.red[0]; imin = s.redi[0];
    __syncthreads();
}

// ---- closed-form H^-1 inner products of constraint rows (row 0 = stability, 1..C = ZMP, C+1..C+F = kinematic)
__device__ __forceinline__ double mdot(const Shared& s, int i, int k)   // M_i . M_k over the footstep columns 1..F
{
    const int a1 = s.k1[i - 1], b1 = s.k1[k - 1];
    const double aw1 = s.w1[i - 1], aw2 = s.w2[i - 1], bw1 = s.w1[k - 1], bw2 = s.w2[k - 1];
    double acc = 0.0;
    // entries: (a1 -> aw1), (a1+1 -> aw2) ; column 0 is the current footstep (not a variable)
    if (a1 >= 1) { if (a1 == b1) acc += aw1 * bw1; else if (a1 == b1 + 1) acc += aw1 * bw2; }
    { const int c = a1 + 1; if (c == b1 && b1 >= 1) acc += aw2 * bw1; else if (c == b1 + 1) acc += aw2 * bw2; }
    return acc;
}
__device__ __forceinline__ double mcol(const Shared& s, int i, int r)   // M_i[r], r in 1..F (0 outside)
{
    if (r < 1) return 0.0;
    const int a1 = s.k1[i - 1];
    if (r == a1) return s.w1[i - 1];
    if (r == a1 + 1) return s.w2[i - 1];
    return 0.0;
}
__device__ __forceinline__ double ip_rows(const Shared& s, const DevA& c, int r1, int r2)
{
    if (r1 > r2) { const int t_ = r1; r1 = r2; r2 = t_; }
    const int C = c.C;
    if (r1 == 0) {
        if (r2 == 0) return c.aa;
        if (r2 <= C) return c.dt * s.PA[r2];
        return 0.0;
    }
    if (r2 <= C) return c.dt * c.dt * (double)r1 + mdot(s, r1, r2) / c.Qf;       // min(r1, r2) = r1
    if (r1 <= C) { const int r = r2 - C; return (-mcol(s, r1, r) + mcol(s, r1, r - 1)) / c.Qf; }
    const int ra = r1 - C, rb = r2 - C;
    if (ra == rb) return (1.0 + (ra >= 2 ? 1.0 : 0.0)) / c.Qf;
    return (rb - ra == 1) ? -1.0 / c.Qf : 0.0;
}

// x += H^-1 N coef : adds  sum_j coef_j * (row_j)  scaled by H^-1 to (u, f).  coef[j] for j < q in s.dp (signed,
// already multiplied by the row's sign); optional extra row `xrow` with coefficient xc.  Result in s.zu / s.zf.
__device__ __forceinline__ void build_direction(Shared& s, const DevA& c, int tid, int q, int xrow, double xc)
{
    const int C = c.C, F = c.F;
    if (tid < C) s.imp[tid] = 0.0;
    __syncthreads();
    // ZMP rows: dt on u[0..i-1]  ->  impulse at i-1, suffix-summed below (a row is active at most once)
    double fpart[MAXF + 1];
#pragma unroll
    for (int k = 0; k <= MAXF; ++k) fpart[k] = 0.0;
    double ce = 0.0;
    for (int j = tid; j <= q; j += T) {
        int row; double cf;
        if (j < q) { row = s.act_row[j]; cf = s.dp[j]; } else { row = xrow; cf = xc; }
        if (row < 0 || cf == 0.0) continue;
        if (row == 0) ce += cf;
        else if (row <= C) {
            s.imp[row - 1] += cf * c.dt;
            const int a1 = s.k1[row - 1];
            if (a1 >= 1) fpart[a1] -= cf * s.w1[row - 1];
            if (a1 + 1 <= F) fpart[a1 + 1] -= cf * s.w2[row - 1];
        } else {
            const int r = row - C;
            fpart[r] += cf;
            if (r >= 2) fpart[r - 1] -= cf;
        }
    }
    // note: two different active ZMP rows never share an index, and the extra row is not active: no write race
    // stability coefficient and footstep parts: one wave scan each, ONE barrier, fixed-order combine (bit reproducible)
    {
        const int lane = tid & 63, wave = tid >> 6;
        const double pe = wave_scan_up(ce);
        if (lane == 63) s.zfpart[wave][0] = pe;
        for (int k = 1; k <= F; ++k) {
            const double pk = wave_scan_up(fpart[k]);
            if (lane == 63) s.zfpart[wave][k] = pk;
        }
    }
    __syncthreads();
    const double cetot = ((s.zfpart[0][0] + s.zfpart[1][0]) + s.zfpart[2][0]) + s.zfpart[3][0];
    if (tid >= 1 && tid <= F) s.zf[tid] = (((s.zfpart[0][tid] + s.zfpart[1][tid]) + s.zfpart[2][tid]) + s.zfpart[3][tid]) / c.Qf;
    // suffix sum of the impulses = total - exclusive prefix
    const double v = (tid < C) ? s.imp[tid] : 0.0;
    double tot;
    const double incl = block_scan_incl_tot(s, v, tid, tot);
    if (tid < C) s.zu[tid] = (tot - (incl - v)) + cetot * s.a[tid];
    __syncthreads();
}

// value of constraint rows for the current x: thread tid < C gets zeta_{tid+1}, threads C..C+F-1 get kin_{tid-C+1}
__device__ __forceinline__ double row_value(Shared& s, const DevA& c, int tid)
{
    const int C = c.C, F = c.F;
    const double cum = block_scan_incl(s, (tid < C) ? s.u[tid] : 0.0, tid);
    if (tid < C) {
        const int a1 = s.k1[tid];
        double mf = 0.0;
        if (a1 >= 1) mf += s.w1[tid] * s.f[a1];
        if (a1 + 1 <= F) mf += s.w2[tid] * s.f[a1 + 1];
        return c.dt * cum - mf;
    }
    if (tid < C + F) { const int r = tid - C + 1; return s.f[r] - (r >= 2 ? s.f[r - 1] : 0.0); }
    return 0.0;
}

__global__ __launch_bounds__(T)
void ismpc_a_tick_kernel(const DevA c, const ismpc_a_state* __restrict__ state_in, ismpc_a_state* __restrict__ state,
                         const double* __restrict__ push, ismpc_a_out* __restrict__ out, int batch)
{
    __shared__ Shared s;
    extern __shared__ double sinv_lds[];            // ldq x ldq when the launch asked for it (c.sinv_in_lds)
    const int tid = threadIdx.x;
    const int C = c.C, F = c.F, P = c.P;
    double* Sinv = c.sinv_in_lds ? sinv_lds : c.scratch + (size_t)blockIdx.x * c.ldq * c.ldq;
    const int ldq = c.ldq;

    for (int work = blockIdx.x; work < 2 * batch; work += gridDim.x) {
        const int inst = work >> 1, axis = work & 1;
        // the two axes of an instance are separate work items: both read the PREVIOUS state (state_in, a copy
        // made by the host entry point) and each writes only its own fields of `state`
        const ismpc_a_state st = state_in[inst];
        const double pos = axis == 0 ? st.x : st.y;
        const double vel = (axis == 0 ? st.xd : st.yd) + (push ? push[inst * 2 + axis] : 0.0);
        const double zmp = axis == 0 ? st.xz : st.yz;
        const double cur = axis == 0 ? st.cur_x : st.cur_y;
        const double off = axis == 0 ? st.off_x : st.off_y;
        const int j = st.j, fc = st.fc;
        const double* fs = axis == 0 ? c.fsx : c.fsy;
        const double* cl = st.rebuilt ? (axis == 0 ? c.clx1 : c.cly1) : (axis == 0 ? c.clx0 : c.cly0);
        const double cloff = st.rebuilt ? off : 0.0;
        int status = 0;
        // ---- validity of indices: fs_plan(fc+1 .. fc+F), cl(j+C+1 .. j+P), j inside step fc
        if (fc < 1 || fc + F > c.n_gait || j < 1 || j + P > c.ncl || j < c.step * (fc - 1) || j > c.step * fc - 1)
            status |= ISMPC_A_ST_BAD_INDEX;

        // ---- mapping (quad_walk_no_plots.m:153-171), bounds (:173-181), stability data
        if (tid < C) {
            const int i = tid + 1;
            int pf = (j + i) / c.step - fc + 1; if (pf < 0) pf = 0;
            const int rem = c.step * (fc + pf) - (j + i);
            double w1, w2;
            if (rem > c.ds) { w1 = 1.0; w2 = 0.0; } else { w1 = (double)rem / c.ds; w2 = 1.0 - (double)rem / c.ds; }
            s.k1[tid] = pf; s.w1[tid] = w1; s.w2[tid] = w2;
            const double m1 = (pf == 0) ? w1 : 0.0;
            s.zhi[tid] = 1.0 * (-zmp + c.w / 2) + m1 * cur;
            s.zlo[tid] = -(-1.0 * (-zmp - c.w / 2) - m1 * cur);
            s.a[tid] = c.a[tid]; s.u[tid] = 0.0;
            s.red[tid] = (pf > F || (w2 != 0.0 && pf + 1 > F) || (rem <= c.ds && pf + 1 > F)) ? 1.0 : 0.0;
        } else s.red[tid] = 0.0;
        for (int k = tid; k <= C; k += T) s.PA[k] = c.PA[k];
        for (int k = tid; k < C + F + 1; k += T) s.state[k] = 0;
        __syncthreads();
        const double ovf = block_sum(s, s.red[tid], tid);
        if (ovf > 0.0) status |= ISMPC_A_ST_OVERFLOW;
        // anticipative tail (:227-231), xfs_store(fsCounter) == current footstep
        double tl = 0.0;
        if (!(status & ISMPC_A_ST_BAD_INDEX))
            for (int i = C + 1 + tid; i <= P; i += T) tl += c.wtail[i - (C + 1)] * ((cl[j + i - 1] + cloff) - cur);
        double tail = block_sum(s, tl, tid);
        if (!(status & ISMPC_A_ST_BAD_INDEX)) tail += c.wP * ((cl[P - 1] + cloff) - cur);
        if (tid == 0) {
            s.b = pos + vel / c.eta - zmp - tail;
            for (int r = 1; r <= F; ++r) {
                double bup = axis == 0 ? c.disp_forw : (c.disp_L / 2 + c.disp_L / 2);
                if (fc == 1 && r == 1) bup = axis == 0 ? c.disp_forw_dummy : (c.disp_L / 2 + c.disp_L / 2);
                double blo = bup;
                if (r == 1) { bup = bup + cur; blo = blo - cur; }
                s.khi[r] = bup; s.klo[r] = -blo;
                const double pr = (status & ISMPC_A_ST_BAD_INDEX) ? 0.0 : fs[fc + r - 1] + off;
                s.pref[r] = pr; s.f[r] = pr;                       // unconstrained minimiser: u = 0, f = p
            }
            s.q = 0; s.iters = 0; s.status = status;
        }
        __syncthreads();

        int q = 0, iters = 0;
        if (status == 0) {
            // ---- equality first: n = (a, 0); from x = (0, p): t = b / a'a
            {
                const double t0 = s.b / c.aa;
                if (tid < C) s.u[tid] = t0 * s.a[tid];
                if (tid == 0) { s.act_row[0] = 0; s.act_sgn[0] = 1.0; s.mu[0] = t0; Sinv[0] = 1.0 / c.aa; }
                q = 1;
                __syncthreads();
            }
            bool resumed = false;
            for (;;) {
                // ======== outer: most violated inactive row (normalised by its H^-1 norm) ========
                const double v = row_value(s, c, tid);
                double cand = INFINITY; int cidx = 0;
                if (tid < C + F) {
                    const int row = tid + 1;
                    if (s.state[row] == 0) {
                        const double lo = tid < C ? s.zlo[tid] : s.klo[tid - C + 1];
                        const double hi = tid < C ? s.zhi[tid] : s.khi[tid - C + 1];
                        const double vl = v - lo, vh = hi - v;
                        const double tol = 1e-11 * (fabs(v) + fmax(fabs(lo), fabs(hi))) + 1e-13;
                        const double nrm = sqrt(ip_rows(s, c, row, row));
                        if (vl < -tol) { cand = vl / nrm; cidx = 2 * row; }
                        if (vh < -tol && vh / nrm < cand) { cand = vh / nrm; cidx = 2 * row + 1; }
                    }
                }
                double vmin; int imin;
                block_argmin(s, cand, cidx, tid, vmin, imin);
                if (!(vmin < 0.0)) {
                    // ---- converged on this working set: two refinement passes (N'x = bounds exactly), then re-check
                    if (resumed) break;
                    for (int pass = 0; pass < 2; ++pass) {
                        const double vv = row_value(s, c, tid);
                        if (tid < C + F && s.state[tid + 1] != 0) s.red[tid] = vv;
                        __syncthreads();
                        // residual per active row (signed), then dm = S^-1 res
                        if (tid < q) {
                            const int row = s.act_row[tid];
                            double res;
                            if (row == 0) {
                                res = 0.0;      // filled below by the block (needs a'u)
                            } else {
                                const double sgn = s.act_sgn[tid];
                                const double bound = row <= C ? (sgn > 0 ? s.zlo[row - 1] : s.zhi[row - 1])
                                                              : (sgn > 0 ? s.klo[row - C] : s.khi[row - C]);
                                res = sgn * (bound - s.red[row - 1]);
                            }
                            s.r[tid] = res;
                        }
                        const double au = block_sum(s, (tid < C) ? s.a[tid] * s.u[tid] : 0.0, tid);
                        if (tid == 0) s.r[0] = s.b - au;
                        __syncthreads();
                        if (tid < q) {
                            double acc = 0.0;
#pragma unroll 8
                            for (int k = 0; k < q; ++k) acc += Sinv[(size_t)k * ldq + tid] * s.r[k];
                            s.dp[tid] = acc * s.act_sgn[tid];
                        }
                        __syncthreads();
                        build_direction(s, c, tid, q, -1, 0.0);
                        if (tid < C) s.u[tid] += s.zu[tid];
                        if (tid >= 1 && tid <= F) s.f[tid] += s.zf[tid];
                        __syncthreads();
                    }
                    resumed = true;
                    continue;                                   // one more feasibility sweep
                }
                resumed = false;
                const int row = imin >> 1;
                const double sg = (imin & 1) ? -1.0 : 1.0;
                double sviol;
                {
                    const int rt = row - 1;                      // thread that holds this row's value
                    if (tid == rt) {
                        const double lo = rt < C ? s.zlo[rt] : s.klo[rt - C + 1];
                        const double hi = rt < C ? s.zhi[rt] : s.khi[rt - C + 1];
                        s.sviol = sg > 0 ? v - lo : hi - v;
                    }
                    __syncthreads();
                    sviol = s.sviol;
                }
                double mu_p = 0.0;
                const double npn = ip_rows(s, c, row, row);
                // ======== inner: steps until the row is added (Goldfarb-Idnani step logic) ========
                for (;;) {
                    if (++iters > c.max_iter) { status |= ISMPC_A_ST_ITER_LIMIT; break; }
                    // d = N' H^-1 n+
                    if (tid < q) s.dp[tid] = sg * s.act_sgn[tid] * ip_rows(s, c, row, s.act_row[tid]);
                    __syncthreads();
                    // r = S^-1 d
                    double racc = 0.0;
                    if (tid < q) {
#pragma unroll 8
                        for (int k = 0; k < q; ++k) racc += Sinv[(size_t)k * ldq + tid] * s.dp[k];
                        s.r[tid] = racc;
                    }
                    const double dr = block_sum(s, (tid < q) ? s.dp[tid] * racc : 0.0, tid);
                    const double gamma = npn - dr;
                    // dual step length: min over active inequalities with r > 0 of mu / r
                    double tc = INFINITY;
                    if (tid >= 1 && tid < q && racc > 0.0) tc = s.mu[tid] / racc;
                    double t1; int l;
                    block_argmin(s, tc, tid, tid, t1, l);
                    const double t2 = (gamma > 1e-12 * npn) ? -sviol / gamma : INFINITY;
                    const double t = fmin(t1, t2);
                    if (!(t < INFINITY)) { status |= (axis == 0 ? ISMPC_A_ST_X_INFEASIBLE : ISMPC_A_ST_Y_INFEASIBLE); break; }
                    if (t2 < INFINITY) {
                        // z = H^-1 (n+ - N r): coefficients -r_j sign_j on the active rows, +sg on the new one
                        if (tid < q) s.dp[tid] = -racc * s.act_sgn[tid];
                        __syncthreads();
                        build_direction(s, c, tid, q, row, sg);
                        if (tid < C) s.u[tid] += t * s.zu[tid];
                        if (tid >= 1 && tid <= F) s.f[tid] += t * s.zf[tid];
                    }
                    if (tid < q) s.mu[tid] -= t * racc;
                    mu_p += t;
                    __syncthreads();
                    if (t2 < INFINITY && t == t2) {
                        // ---- full step: border update of S^-1, append the row
                        const double ig = 1.0 / gamma;
                        if (tid < q) {
                            const double rj = s.r[tid];
#pragma unroll 8
                            for (int k = 0; k < q; ++k) Sinv[(size_t)k * ldq + tid] += s.r[k] * rj * ig;
                            Sinv[(size_t)q * ldq + tid] = -rj * ig;
                            Sinv[(size_t)tid * ldq + q] = -rj * ig;
                        }
                        if (tid == 0) {
                            Sinv[(size_t)q * ldq + q] = ig;
                            s.act_row[q] = row; s.act_sgn[q] = sg; s.mu[q] = mu_p; s.state[row] = sg > 0 ? 1 : -1;
                        }
                        ++q;
                        __syncthreads();
                        break;
                    }
                    // ---- partial step: drop working-set entry l (Schur update), keep going with the same row
                    {
                        const double piv = Sinv[(size_t)l * ldq + l];
                        __syncthreads();
                        if (tid < q) s.r[tid] = Sinv[(size_t)l * ldq + tid];      // column l (symmetric)
                        __syncthreads();
                        if (tid < q && tid != l) {
                            const double cj = s.r[tid] / piv;
#pragma unroll 8
                            for (int k = 0; k < q; ++k) if (k != l) Sinv[(size_t)k * ldq + tid] -= s.r[k] * cj;
                        }
                        __syncthreads();
                        // move the last entry into slot l
                        const int last = q - 1;
                        if (l != last) {
                            if (tid < q && tid != l) {
                                const double vlast = Sinv[(size_t)last * ldq + tid];
                                Sinv[(size_t)l * ldq + tid] = vlast;
                                Sinv[(size_t)tid * ldq + l] = vlast;
                            }
                            __syncthreads();
                            if (tid == 0) Sinv[(size_t)l * ldq + l] = Sinv[(size_t)last * ldq + last];
                        }
                        if (tid == 0) {
                            s.state[s.act_row[l]] = 0;
                            if (l != last) { s.act_row[l] = s.act_row[last]; s.act_sgn[l] = s.act_sgn[last]; s.mu[l] = s.mu[last]; }
                        }
                        --q;
                        __syncthreads();
                    }
                    // violation of the row at the new point
                    {
                        const double vv = row_value(s, c, tid);
                        const int rt = row - 1;
                        if (tid == rt) {
                            const double lo = rt < C ? s.zlo[rt] : s.klo[rt - C + 1];
                            const double hi = rt < C ? s.zhi[rt] : s.khi[rt - C + 1];
                            s.sviol = sg > 0 ? vv - lo : hi - vv;
                        }
                        __syncthreads();
                        sviol = s.sviol;
                    }
                }
                if (status != 0) break;
            }
        }

        // ---- LIP update (:297-322), footstep bookkeeping (:522-556), outputs
        __syncthreads();
        if (tid == 0) {
            const double u0 = (status & (ISMPC_A_ST_BAD_INDEX | ISMPC_A_ST_OVERFLOW)) ? 0.0 : s.u[0];
            const double f0 = (status & (ISMPC_A_ST_BAD_INDEX | ISMPC_A_ST_OVERFLOW)) ? cur : s.f[1];
            const double p0 = pos, v0 = vel, z0 = zmp;
            const double np_ = (c.Au[0] * p0 + c.Au[1] * v0 + c.Au[2] * z0) + c.Bu[0] * u0;
            const double nv_ = (c.Au[3] * p0 + c.Au[4] * v0 + c.Au[5] * z0) + c.Bu[1] * u0;
            const double nz_ = (c.Au[6] * p0 + c.Au[7] * v0 + c.Au[8] * z0) + c.Bu[2] * u0;
            ismpc_a_state* so = state + inst;
            const bool ok = (status & (ISMPC_A_ST_BAD_INDEX | ISMPC_A_ST_OVERFLOW)) == 0;
            const bool stepped = ok && (j + 1 >= c.step * fc);
            if (ok) {
                if (axis == 0) { so->x = np_; so->xd = nv_; so->xz = nz_; } else { so->y = np_; so->yd = nv_; so->yz = nz_; }
                if (stepped) {
                    const double noff = f0 - fs[fc];                  // predicted - fs_plan(fc+1)  (base plan)
                    if (axis == 0) { so->cur_x = f0; so->off_x = noff; } else { so->cur_y = f0; so->off_y = noff; }
                }
                if (axis == 0) { so->j = j + 1; if (stepped) { so->fc = fc + 1; so->rebuilt = 1; } }
            }
            if (out) {
                ismpc_a_out* o = out + inst;
                o->com_before[axis] = pos; o->vel_after[axis] = ok ? nv_ : vel; o->u0[axis] = u0; o->f0[axis] = f0;
                if (axis == 0) { o->iters_x = iters; atomicOr(&o->status, status); atomicOr(&o->active, q & 0xffff); }
                else { o->iters_y = iters; atomicOr(&o->status, status); atomicOr(&o->active, (q & 0xffff) << 16); }
            }
        }
        __syncthreads();
    }
}

// ---- swing-foot re-placement: one thread per instance (closed forms; the 2-/4-variable quadprog is separable, so its
// minimiser is the projection of the target on the box).  trotting/quad_as_bip_no_plots.m:332-426 + compute_two_feet1.m,
// walking/quad_walk_no_plots.m:336-504 + compute_one_feet_walk.m:84-140.
struct FeetParams { int gait, rows; double phi, disp_i, disp_o, disp_forw; };
struct FeetParamsSet { FeetParams p[4]; };           // per base plan (ismpc_a_inst.plan): Monte-Carlo batches mix trot and walk instances

__device__ __forceinline__ void fixed_diagonal(double fx1, double fy1, double fx2, double fy2, double zx, double zy,
                                               double& m, double& dx, double& dy)
{
    m = (fy2 - fy1) / (fx2 - fx1);
    const double q = fy1 - m * fx1;
    const double xi = (zy + m * zx - q) / (2 * m), yi = m * xi + q;
    dx = zx - xi; dy = zy - yi;
}
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void ismpc_a_feet_kernel(const FeetParamsSet fset, const ismpc_a_inst* __restrict__ inst, int nplans, const ismpc_a_state* __restrict__ prev,
                                    const ismpc_a_out* __restrict__ out, double* __restrict__ feet, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    if (out[b].status & (ISMPC_A_ST_BAD_INDEX | ISMPC_A_ST_OVERFLOW)) return;
    int pl = inst ? inst[b].plan : 0;                            // per-instance gait parameters: the foot rules of the instance's base plan
    if (pl < 0 || pl >= nplans) return;
    const FeetParams fpz = fset.p[pl];
    const int fc = prev[b].fc;                                   // the fsCounter this tick ran with
    if (fc < 1 || fc + 8 >= fpz.rows) return;
    double* fp = feet + (size_t)b * fpz.rows * 8;
#define FPL(r, c) fp[(size_t)((r) - 1) * 8 + ((c) - 1)]
    const double zx = out[b].f0[0], zy = out[b].f0[1];           // predicted_xfs(1), predicted_yfs(1)
    const double di = fpz.disp_i, dob = fpz.disp_o, df = fpz.disp_forw;
    if (fpz.gait == 0) {
        const bool odd = (fc % 2) == 1;
        const int f1 = odd ? 3 : 1, f2 = odd ? 7 : 5, m1 = odd ? 1 : 3, m2 = odd ? 5 : 7;
        double m, dx, dy;
        fixed_diagonal(FPL(fc, f1), FPL(fc, f1 + 1), FPL(fc, f2), FPL(fc, f2 + 1), zx, zy, m, dx, dy);
        const double a1x = FPL(fc + 1, m1), a1y = FPL(fc + 1, m1 + 1), a2x = FPL(fc + 1, m2), a2y = FPL(fc + 1, m2 + 1);
        double x1, y1, x2, y2;
        if (fpz.phi == 3.14159265358979323846 / 2) {
            x1 = a1x; x2 = a2x; y1 = zy - m * (x1 - zx); y2 = zy - m * (x2 - zx);
        } else {
            const double tp = tan(fpz.phi);
            x1 = (zy + m * zx - a1y + tp * a1x) / (tp + m); y1 = tp * (x1 - a1x) + a1y;
            x2 = (zy + m * zx - a2y + tp * a2x) / (tp + m); y2 = tp * (x2 - a2x) + a2y;
        }
        if (dy != 0 || dx != 0) {
            FPL(fc + 1, m1) = x1; FPL(fc + 1, m1 + 1) = y1; FPL(fc + 1, m2) = x2; FPL(fc + 1, m2 + 1) = y2;
            FPL(fc + 1, f1) = FPL(fc, f1); FPL(fc + 1, f1 + 1) = FPL(fc, f1 + 1); FPL(fc + 1, f2) = FPL(fc, f2); FPL(fc + 1, f2 + 1) = FPL(fc, f2 + 1);
        }
        const double lo_ = (fc == 1) ? dob / 2 : dob, li_ = (fc == 1) ? di / 2 : di, lf_ = (fc == 1) ? df / 2 : df;
        { const double px = FPL(fc, m1), py = FPL(fc, m1 + 1);
          FPL(fc + 1, m1 + 1) = clipd(FPL(fc + 1, m1 + 1), py - li_, py + lo_);
          if (FPL(fc + 1, m1) > px + lf_) FPL(fc + 1, m1) = px + lf_; }
        { const double px = FPL(fc, m2), py = FPL(fc, m2 + 1);
          FPL(fc + 1, m2 + 1) = clipd(FPL(fc + 1, m2 + 1), py - lo_, py + li_);
          if (FPL(fc + 1, m2) > px + lf_) FPL(fc + 1, m2) = px + lf_; }
    } else {
        const int counter = fc;                                  // `counter` (quad_walk_no_plots.m:114,527) starts at 1 and moves with fsCounter
        if (!(counter == 2 || counter == 4 || counter == 6 || counter == 8)) return;
        int mc, a1, a2; bool outer_up;
        if (counter == 2)      { mc = 7; a1 = 1; a2 = 5; outer_up = true; }
        else if (counter == 4) { mc = 3; a1 = 1; a2 = 5; outer_up = false; }
        else if (counter == 6) { mc = 5; a1 = 3; a2 = 7; outer_up = false; }
        else                   { mc = 1; a1 = 3; a2 = 7; outer_up = true; }
        double m, dx, dy;
        fixed_diagonal(FPL(fc, a1), FPL(fc, a1 + 1), FPL(fc, a2), FPL(fc, a2 + 1), zx, zy, m, dx, dy);
        const double xfree = FPL(fc + 1, mc) + dx, yfree = FPL(fc + 1, mc + 1) + dy;
        if (dy != 0 || dx != 0)
            for (int l = 1; l <= 8; ++l) { FPL(fc + l, mc) = xfree; FPL(fc + l, mc + 1) = yfree; }
        const bool dummy = (counter == 2 || counter == 4) && fc <= 4;
        const double lo_ = dummy ? dob / 2 : dob, li_ = dummy ? di / 2 : di, lf_ = dummy ? df / 2 : df;
        const double px = FPL(fc, mc), py = FPL(fc, mc + 1);
        double X1 = FPL(fc + 1, mc), X2 = FPL(fc + 1, mc + 1);
        X2 = outer_up ? clipd(X2, py - li_, py + lo_) : clipd(X2, py - lo_, py + li_);
        if (X1 > px + lf_) X1 = px + lf_;
        if (counter == 8) { for (int l = 1; l <= 8; ++l) FPL(fc + l, mc) = X1; FPL(fc + 1, mc + 1) = X2; }    // :498-503 as written
        else for (int l = 1; l <= 8; ++l) { FPL(fc + l, mc) = X1; FPL(fc + l, mc + 1) = X2; }
    }
#undef FPL
}

__global__ void ismpc_a_feet_fill(const double* __restrict__ base, double* __restrict__ feet, int rows, int batch)
{
    const size_t n = (size_t)batch * rows * 8;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) feet[e] = base[e % ((size_t)rows * 8)];
}
// every instance starts from the foot plan of ITS base plan (base: nplans x rows x 8)
__global__ void ismpc_a_feet_fill_inst(const double* __restrict__ base, const ismpc_a_inst* __restrict__ inst, int nplans, double* __restrict__ feet, int rows, int batch)
{
    const size_t per = (size_t)rows * 8, n = (size_t)batch * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / per;
        int pl = inst[b].plan; if (pl < 0 || pl >= nplans) pl = 0;
        feet[e] = base[(size_t)pl * per + (e - b * per)];
    }
}

// Per-instance gait parameters: the instances of a batch differ in their footstep count F_i (3..6), and a QP costs what the
// kernel instantiated for its F costs (border of 2F+1 columns, F(F+1)/2 + 2F + 2 Gram sums per block solve).  The instances
// are therefore listed by F_i (order of arrival inside a list is irrelevant: QPs are independent) and each list runs through
// the kernel of its own shape; records the kernel would reject (F out of range ...) go with F = 3 and are flagged there.
__global__ void ismpc_a_bucket_by_F(const ismpc_a_inst* __restrict__ inst, int batch, int Fmax, int* __restrict__ order, int cap, int* __restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    int f = inst[i].F;
    if (f < 3) f = 3;
    if (f > Fmax) f = 3;
    const int b = f - 3;
    order[(size_t)b * cap + atomicAdd(&counts[b], 1)] = i;
}

// Everything a tick needs before its solver launch, in one launch instead of a copy, a clear and a memset: the snapshot of the
// state the two QPs of an instance read (the solver updates `state` in place), cleared flags of the output records, zeroed
// work counters.  96-byte state records move as six 16-byte words per thread.
// With per-instance gait parameters also the PiPre record of every instance (ismpc_a_dev.hpp).
__global__ void ismpc_a_tick_prologue(const ismpc_a_state* __restrict__ state, ismpc_a_state* __restrict__ prev, ismpc_a_out* out, int batch, int* counters,
                                      const ismpc_a_inst* __restrict__ inst, ismpc_a::PiPre* __restrict__ pre, double grav, double dt, int C, int P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 && counters) counters[i] = 0;
    // the snapshot as one flat, coalesced copy of 16-byte words (a thread per record moved six words 96 bytes apart per wavefront access)
    static_assert(sizeof(ismpc_a_state) % 16 == 0, "state record: whole 16-byte words");
    {
        constexpr int WPR = (int)(sizeof(ismpc_a_state) / 16);
        const double2* src = reinterpret_cast<const double2*>(state);
        double2* dst = reinterpret_cast<double2*>(prev);
        const long long nw = (long long)batch * WPR;
        for (long long w = i; w < nw; w += (long long)gridDim.x * blockDim.x) dst[w] = src[w];
    }
    if (i >= batch) return;
    if (inst && pre) {
        const double height = inst[i].height;
        const double eta = (height > 0) ? sqrt(grav / height) : 1.0;      // (a record the solver rejects: any finite values)
        const double lam = exp(-eta * dt);
        auto ipw = [](double b, int n) { double r = 1.0; while (n > 0) { if (n & 1) r *= b; b *= b; n >>= 1; } return r; };
        const double lamC = ipw(lam, C), lamP = ipw(lam, P);
        const double r1 = 1.0 / (1.0 - lam);
        const double k1c = (1 / eta) * (1 - lam) / (1 - lamC), k2c = dt * 1.0 * lamC;
        ismpc_a::PiPre q;
        q.eta = eta; q.lam = lam; q.lamC = lamC; q.lamP = lamP; q.k1c = k1c; q.k2c = k2c;
        q.A1 = k1c * r1; q.A2 = k1c * k1c / ((1.0 - lam) * (1.0 + lam)); q.B2 = 2.0 * k1c * k2c * r1;
        q.aa = (q.A2 * ((1.0 - lamC) * (1.0 + lamC)) - q.B2 * (1.0 - lamC)) + (double)C * (k2c * k2c);     // = the kernel's sum_{k<C} a_k^2
        const double Qf = (inst[i].Qf > 0) ? inst[i].Qf : 1.0;
        const int step = inst[i].step >= 2 ? inst[i].step : 2, ds = inst[i].ds >= 2 ? inst[i].ds : 2;
        q.sqQf = sqrt(Qf); q.isqQf = 1.0 / sqrt(Qf); q.iQf = 1.0 / Qf; q.ieta = 1.0 / eta;
        q.inv_ds = 1.0 / (double)ds; q.inv_dsm1 = 1.0 / (double)(ds - 1); q.rstep = 1.0f / (float)step; q.pad_ = 0;
        const double ie = 1.0 / lam;
        q.ch = 0.5 * (ie + lam); q.sh = 0.5 * (ie - lam); q.sh_eta = q.sh / eta;
        pre[i] = q;
    }
    if (out) { out[i].status = 0; out[i].active = 0; out[i].iters_x = 0; out[i].iters_y = 0; }
}

struct DeviceGuardA {          // entry points leave the caller's current device as they found it
    int prev = -1, dev; hipError_t err = hipSuccess;
    explicit DeviceGuardA(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceGuardA() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
thread_local std::string g_err_a = "";
int fail_a(int code, const std::string& msg) { g_err_a = msg; return code; }
#define HIP_TRY_A(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail_a(-2, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define ON_DEVICE_A(h_) DeviceGuardA guard_((h_)->device); HIP_TRY_A(guard_.err)

// MATLAB linspace(d1, d2, n)
void linspace_m(double d1, double d2, int n, std::vector<double>& y)
{
    y.resize(n);
    const int n1 = n - 1;
    for (int k = 0; k <= n1; ++k) y[k] = d1 + (k * (d2 - d1)) / n1;
    if (n > 0) { y[0] = d1; y[n1] = d2; }
}
// quad_walk_no_plots.m:86-99 (initial) / :540-549 (rebuilt)
void centreline(const std::vector<double>& fs, int step, int ds, int NF, bool initial, std::vector<double>& cl)
{
    cl.clear();
    std::vector<double> lin;
    if (initial) {
        for (int k = 0; k < step - ds; ++k) cl.push_back(fs[0] * 1.0);
        linspace_m(fs[0], fs[1], ds, lin);
        cl.insert(cl.end(), lin.begin(), lin.end());
    } else {
        for (int k = 0; k < step; ++k) cl.push_back(fs[0] * 1.0);
    }
    for (int i = 2; i <= NF - 1; ++i) {
        for (int k = 0; k < step - ds; ++k) cl.push_back(fs[i - 1] * 1.0);
        linspace_m(fs[i - 1], fs[i], ds, lin);
        cl.insert(cl.end(), lin.begin(), lin.end());
    }
}

}  // namespace

struct ismpc_a_handle {
    ismpc_a_params p{};
    DevA c{};
    int device = 0, slots = 0;
    ismpc_a_state* prev = nullptr; int prev_cap = 0;     // copy of the state the tick reads
    FeetParams feet{}; double* feet_base = nullptr;     // swing-foot QPs (ismpc_a_feet_init_device)
    FeetParamsSet feet_set{}; int feet_plans = 0;        // ... and per base plan (ismpc_a_feet_init_inst_device)
    bool use_wave = true; int wave_blocks = 0;           // structured wavefront-per-QP kernel (default) vs workgroup-per-QP
    int cus = 0, wave_occ[16] = {0};                     // resident workgroups per CU of the wave kernels ([F - 3][precision x per-instance])
    ismpc_a::PiPre* pre = nullptr; int pre_cap = 0;       // per-instance launches: the prologue's record per instance
    int* order = nullptr; int order_cap = 0;             // per-instance launches: instance lists by footstep count (4 x cap) + 4 counters
    bool bucket_by_F = false;                            // ISMPC_A_BUCKET=1: one launch per footstep count instead of one launch of the widest kernel
                                                         // (measured slower: 6.2 vs 4.0 ms at 16 384 instances -- four tails of 100-iteration QPs instead of one)
    int precision = 0;                                   // 0: the QPs are solved in fp64, 1: in fp32 (ismpc_a_set_precision)
    DevA* c_dev = nullptr; bool c_dirty = true;          // the constants in device memory (what the wave kernels read), re-sent after a change
    int* work_counter = nullptr;                          // [0] the launch's counter, [1] the fp64 re-solve's, [2] deferred QPs of the fp32 launch
    int resolve_grid = 64;                                // workgroups of the fp64 re-solve launch behind an fp32 launch (ISMPC_A_RESOLVE_GRID)
    int* defer_list = nullptr; int defer_cap = 0; bool defer_off = false;   // fp32 solve: QPs handed to the fp64 instantiation (ISMPC_A_F32_RESOLVE=0: none)
    unsigned long long* hist = nullptr; int hist_cap = 0;   // per-QP working set of the previous tick (closed-loop first guess)
    int claim_chunk = 0;                                  // 0: by shape (tick_launch), else ISMPC_A_CLAIM
    int static_q = 8;                                     // sixteenths of a launch dealt out without atomics (ISMPC_A_STATIC; scripts/claim_sweep.sh:
                                                          // half is +2-12 % on every bench leg, three quarters starts to cost balance)
    bool hist_ticks = false, hist_valid = false;           // use it in plain tick calls too / it holds the previous tick of this batch
    int hist_batch = 0; bool hist_off = false;            // ISMPC_A_HISTORY=0: never (A/B)
    hipStream_t last_stream = nullptr; bool used = false; // stream of the previous launch: scratch that outlives a call is re-allocated only
                                                          // after that stream has drained (grow_sync)
    std::vector<void*> allocs;
    std::vector<double> fsx, fsy;
};

namespace {
template <typename Tp>
int upload_a(ismpc_a_handle* h, const std::vector<Tp>& v, const Tp** dst)
{
    void* p = nullptr;
    HIP_TRY_A(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(Tp)));
    h->allocs.push_back(p);
    if (!v.empty()) HIP_TRY_A(hipMemcpy(p, v.data(), v.size() * sizeof(Tp), hipMemcpyHostToDevice));
    *dst = static_cast<const Tp*>(p);
    return 0;
}
}  // namespace

extern "C" {

const char* ismpc_a_last_error(void) { return g_err_a.c_str(); }

void ismpc_a_params_default(int gait, ismpc_a_params* p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    if (gait == 1) { p->C = 100; p->P = 200; p->step = 50; p->ds = 30; p->Qf = 1e9; }     // quad_walk_no_plots.m:20-45,271
    else           { p->C = 160; p->P = 320; p->step = 80; p->ds = 50; p->Qf = 1e7; }     // quad_as_bip_no_plots.m:16-39,257
    p->F = 3; p->n_gait = 100; p->dt = 0.01; p->height = 0.56; p->grav = 9.8; p->w = 0.02;
    p->disp_forw = 0.5; p->disp_forw_dummy = 0.25; p->disp_L = 0.4;
}

void ismpc_a_gait_default(int gait, double phi, double disp_A, ismpc_a_gait* g)
{
    if (!g) return;
    g->gait = gait; g->n_gait = 100; g->disp_A = disp_A; g->phi = phi;
    g->disp_B = 0.259394; g->disp_C = 0.88; g->disp_i = 0.4; g->disp_o = 0.4; g->disp_forw = 0.5;
}

// trotting/init_quadruped.m:5-184, walking/init_quadruped2.m:5-284 (host; once per run)
int ismpc_a_plan(const ismpc_a_gait* g, double* foot_plan, double* center)
{
    if (!g || !foot_plan || !center || g->n_gait < 16) return fail_a(-1, "bad argument");
    const int NG = g->n_gait;
    const double dfd = g->disp_forw / 2, dv = std::min(g->disp_i, g->disp_o), dvd = dv / 2;
    double xs = g->disp_A * std::cos(g->phi), ys = g->disp_A * std::sin(g->phi);
    double xsd = g->disp_A * std::cos(g->phi) / 2, ysd = g->disp_A * std::sin(g->phi) / 2;
    auto clip = [&](double& x, double& y, double vlim, double flim) {
        if (y > vlim || x > flim) {
            if (g->phi > std::atan(vlim / flim)) { y = vlim; x = vlim * std::cos(g->phi) / std::sin(g->phi); }
            else { x = flim; y = flim * std::sin(g->phi) / std::cos(g->phi); }
        }
    };
    clip(xsd, ysd, dvd, dfd);          // first (half) step   :62-81
    clip(xs, ys, dv, g->disp_forw);    // regular step        :84-102
    const int rows = NG + 1;           // 1-based rows 1..NG+1 stored at index row-1
    std::vector<double> fp((size_t)(rows + 8) * 8);
    auto FP = [&](int r, int col) -> double& { return fp[(size_t)(r - 1) * 8 + (col - 1)]; };
    for (int r = 1; r <= rows + 7; ++r) {
        FP(r,1) = 0.0; FP(r,2) = g->disp_B; FP(r,3) = 0.0; FP(r,4) = -g->disp_B;
        FP(r,5) = g->disp_C; FP(r,6) = -g->disp_B; FP(r,7) = g->disp_C; FP(r,8) = g->disp_B;
    }
    auto cross = [&](int r, double& cx, double& cy) {      // intersection of the diagonals BL-FR and BR-FL
        const double m1 = (FP(r,6) - FP(r,2)) / (FP(r,5) - FP(r,1)), b1 = FP(r,2) - m1 * FP(r,1);
        const double m2 = (FP(r,8) - FP(r,4)) / (FP(r,7) - FP(r,3)), b2 = FP(r,4) - m2 * FP(r,3);
        cx = (b2 - b1) / (m1 - m2); cy = m1 * cx + b1;
    };
    for (int r = 0; r < NG; ++r) { center[r * 2] = 0.0; center[r * 2 + 1] = 0.0; }
    center[0] = g->disp_C / 2;
    int used = NG;
    if (g->gait == 0) {
        FP(2,1) = xsd; FP(2,5) = g->disp_C + xsd; FP(2,2) = g->disp_B + ysd; FP(2,6) = -g->disp_B + ysd;
        for (int j = 3; j <= NG; ++j) {
            const bool even = (j % 2) == 0;
            const int mv1 = even ? 1 : 3, mv2 = even ? 5 : 7, hd1 = even ? 3 : 1, hd2 = even ? 7 : 5;
            FP(j, mv1) = FP(j-1, mv1) + xs; FP(j, mv2) = FP(j-1, mv2) + xs; FP(j, hd1) = FP(j-1, hd1); FP(j, hd2) = FP(j-1, hd2);
            FP(j, mv1+1) = FP(j-1, mv1+1) + ys; FP(j, mv2+1) = FP(j-1, mv2+1) + ys; FP(j, hd1+1) = FP(j-1, hd1+1); FP(j, hd2+1) = FP(j-1, hd2+1);
        }
        for (int k = 2; k <= NG; ++k) cross(k, center[(k-1)*2], center[(k-1)*2+1]);
    } else {
        FP(3,7) = g->disp_C + xsd; FP(4,7) = FP(3,7); FP(5,7) = FP(3,7);
        FP(2,3) = FP(1,3); FP(3,3) = FP(1,3); FP(4,3) = FP(3,3); FP(5,3) = FP(4,3) + xsd;
        FP(3,8) = g->disp_B + ysd; FP(4,8) = FP(3,8); FP(5,8) = FP(3,8);
        FP(2,4) = FP(1,4); FP(3,4) = FP(1,4); FP(4,4) = FP(3,4); FP(5,4) = FP(4,4) + ysd;
        for (int j = 6; j <= NG; j += 8) {
            for (int cc = 0; cc < 2; ++cc) {
                const double st = cc == 0 ? xs : ys;
                const int BL = 1 + cc, BR = 3 + cc, FR = 5 + cc, FL = 7 + cc;
                FP(j,FR) = FP(j-1,FR); FP(j+1,FR) = FP(j,FR) + st; for (int k = 2; k <= 7; ++k) FP(j+k,FR) = FP(j+1,FR);
                FP(j,BL) = FP(j-1,BL); FP(j+1,BL) = FP(j,BL); FP(j+2,BL) = FP(j,BL); FP(j+3,BL) = FP(j+2,BL) + st;
                for (int k = 4; k <= 7; ++k) FP(j+k,BL) = FP(j+3,BL);
                FP(j,FL) = FP(j-1,FL); for (int k = 1; k <= 4; ++k) FP(j+k,FL) = FP(j,FL);
                FP(j+5,FL) = FP(j+4,FL) + st; FP(j+6,FL) = FP(j+5,FL); FP(j+7,FL) = FP(j+5,FL);
                FP(j,BR) = FP(j-1,BR); for (int k = 1; k <= 6; ++k) FP(j+k,BR) = FP(j,BR);
                FP(j+7,BR) = FP(j+6,BR) + st;
            }
            used = std::max(used, j + 7);
        }
        used = std::min(used, NG + 1);
        for (int j = 1; j <= NG - 4; j += 8) {
            for (int k = 0; k <= 6; k += 2) cross(j + k, center[(j+k-1)*2], center[(j+k-1)*2+1]);
            for (int k = 1; k <= 7; k += 2) { center[(j+k-1)*2] = center[(j+k-2)*2]; center[(j+k-1)*2+1] = center[(j+k-2)*2+1]; }
        }
    }
    std::memcpy(foot_plan, fp.data(), sizeof(double) * (size_t)used * 8);
    return used;
}

int ismpc_a_create(const ismpc_a_params* p, const double* center, int device, ismpc_a_handle** out)
{
    if (!p || !center || !out) return fail_a(-1, "null argument");
    *out = nullptr;
    if (p->C < 2 || p->F < 1 || p->F > MAXF || p->C + p->F > T || p->P <= p->C || p->step < 2 || p->ds < 2 || p->ds >= p->step ||
        p->n_gait < p->F + 2 || !(p->dt > 0) || !(p->height > 0) || !(p->Qf > 0) || !(p->w >= 0))
        return fail_a(-1, "unsupported parameters (need 2 <= C, C + F <= 256, 1 <= F <= 8, P > C, 2 <= ds < step)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail_a(-2, "no HIP device visible: the ISMPC hot path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail_a(-1, "device ordinal out of range");
    ismpc_a_handle* h = new (std::nothrow) ismpc_a_handle();
    if (!h) return fail_a(-3, "out of host memory");
    h->p = *p; h->device = device;
    DeviceGuardA guard_(device);
    if (guard_.err != hipSuccess) { delete h; return fail_a(-2, "hipSetDevice failed"); }
    DevA& c = h->c;
    c.C = p->C; c.P = p->P; c.F = p->F; c.step = p->step; c.ds = p->ds; c.n_gait = p->n_gait;
    c.dt = p->dt; c.eta = std::sqrt(p->grav / p->height); c.w = p->w; c.Qf = p->Qf;
    c.disp_forw = p->disp_forw; c.disp_forw_dummy = p->disp_forw_dummy; c.disp_L = p->disp_L;
    c.ldq = (p->C + p->F + 2) | 1;                        // odd leading dimension: conflict-free LDS columns
    c.max_iter = 20 * (p->C + p->F) + 200;
    if (const char* e = std::getenv("ISMPC_A_HISTORY")) h->hist_off = std::atoi(e) == 0;
    if (const char* e = std::getenv("ISMPC_A_BUCKET")) h->bucket_by_F = std::atoi(e) != 0;
    if (const char* e = std::getenv("ISMPC_A_PRECISION")) h->precision = (!std::strcmp(e, "f32") && p->F >= 3 && p->F <= 6) ? 1 : 0;   // A/B knob
    // ISMPC_A_WARM=add,drop,extra,min_viol,gi_first,peel,rounds,round_adds overrides; ISMPC_A_WARM=0 starts every QP cold
    c.warm_add = 8; c.warm_drop = 12; c.warm_extra = 0; c.warm_min_viol = 6; c.warm_gi = 2; c.warm_peel_end = 1;
    c.warm_rounds = 2; c.warm_round_adds = 8;
    if (const char* e = std::getenv("ISMPC_A_F32_RESOLVE")) h->defer_off = std::atoi(e) == 0;
    if (const char* e = std::getenv("ISMPC_A_RESOLVE_GRID")) h->resolve_grid = std::max(1, std::min(std::atoi(e), 1024));
    if (const char* e = std::getenv("ISMPC_A_STATIC")) h->static_q = std::max(0, std::min(std::atoi(e), 16));
    if (const char* e = std::getenv("ISMPC_A_CLAIM")) h->claim_chunk = std::max(0, std::min(std::atoi(e), 64));   // 0: by shape
    if (const char* e = std::getenv("ISMPC_A_WARM")) {
        int a_ = 0, d_ = 0, x_ = 0, v_ = 0, g_ = 0, pe_ = 0, r_ = 0, ra_ = 0;
        const int got = std::sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d", &a_, &d_, &x_, &v_, &g_, &pe_, &r_, &ra_);
        if (got >= 7) c.warm_rounds = std::max(0, std::min(r_, 16));
        if (got >= 8) c.warm_round_adds = std::max(1, std::min(ra_, 64));
        if (got >= 5) c.warm_gi = std::max(0, std::min(g_, 64));
        if (got >= 6) c.warm_peel_end = pe_ != 0;
        if (got >= 1) c.warm_add = std::max(0, std::min(a_, 32));
        if (got >= 2) c.warm_drop = std::max(0, std::min(d_, 32));
        if (got >= 3) c.warm_extra = std::max(0, std::min(x_, 8));
        if (got >= 4) c.warm_min_viol = std::max(1, std::min(v_, 256));
    }
    // S^-1 lives in an L2-resident scratch slab (4 workgroups per CU); ISMPC_A_SINV=lds keeps it in LDS instead when it
    // fits next to the static block (then 1 workgroup per CU).  Measured on MI355X (walk, C=100, batch 16 384):
    // scratch 2.8e5 ticks/s, LDS 2.0e5 ticks/s -- the kernel is barrier-latency bound, concurrency wins.
    c.sinv_in_lds = 0;
    if (const char* e = std::getenv("ISMPC_A_SINV")) {
        if (!std::strcmp(e, "lds") && (size_t)c.ldq * c.ldq * sizeof(double) + sizeof(Shared) + 1024 <= 160u * 1024u) c.sinv_in_lds = 1;
    }
    const double eta = c.eta, dt = c.dt;
    const double ch = std::cosh(eta * dt), sh = std::sinh(eta * dt);                           // :67-71
    const double Au[9] = { ch, sh / eta, 1 - ch, eta * sh, ch, -eta * sh, 0, 0, 1 };
    const double Bu[3] = { dt - sh / eta, 1 - ch, dt };
    std::memcpy(c.Au, Au, sizeof(Au)); std::memcpy(c.Bu, Bu, sizeof(Bu));
    // stability row (:233-238) and tail weights (:229-231)
    const double lambda = std::exp(-eta * dt);
    std::vector<double> a(p->C), PA(p->C + 1, 0.0), PA2(p->C + 1, 0.0), wt(p->P - p->C);
    double aa = 0.0;
    for (int i = 0; i < p->C; ++i) {
        a[i] = (1 / eta) * (1 - lambda) / (1 - std::pow(lambda, p->C)) * std::exp(-eta * dt * i) - dt * 1.0 * std::exp(-eta * dt * p->C);
        PA[i + 1] = PA[i] + a[i]; aa += a[i] * a[i]; PA2[i + 1] = aa;
    }
    double sumw = 0.0;
    for (int i = p->C + 1; i <= p->P; ++i) { wt[i - (p->C + 1)] = std::exp(-eta * dt * i) * (1 - std::exp(-eta * dt)); sumw += wt[i - (p->C + 1)]; }
    c.wP = std::exp(-eta * dt * p->P); c.sumw = sumw + c.wP; c.aa = aa;
    c.sqQf = std::sqrt(c.Qf); c.isqQf = 1.0 / std::sqrt(c.Qf); c.iQf = 1.0 / c.Qf; c.ieta = 1.0 / eta;
    c.inv_ds = 1.0 / (double)p->ds; c.rstep = 1.0f / (float)p->step;
    h->fsx.resize(p->n_gait); h->fsy.resize(p->n_gait);
    for (int i = 0; i < p->n_gait; ++i) { h->fsx[i] = center[i * 2]; h->fsy[i] = center[i * 2 + 1]; }
    std::vector<double> clx0, cly0, clx1, cly1;
    centreline(h->fsx, p->step, p->ds, p->n_gait, true, clx0);  centreline(h->fsy, p->step, p->ds, p->n_gait, true, cly0);
    centreline(h->fsx, p->step, p->ds, p->n_gait, false, clx1); centreline(h->fsy, p->step, p->ds, p->n_gait, false, cly1);
    c.ncl = (int)std::min(clx0.size(), clx1.size());
    int rc = upload_a(h, a, &c.a);
    if (!rc) rc = upload_a(h, PA, &c.PA);
    if (!rc) rc = upload_a(h, PA2, &c.PA2);
    if (!rc) rc = upload_a(h, wt, &c.wtail);
    if (!rc) rc = upload_a(h, h->fsx, &c.fsx);
    if (!rc) rc = upload_a(h, h->fsy, &c.fsy);
    if (!rc) { c.plan_x[0] = c.fsx; c.plan_y[0] = c.fsy; c.nplans = 1; c.grav = p->grav; }
    if (!rc) rc = upload_a(h, clx0, &c.clx0);
    if (!rc) rc = upload_a(h, cly0, &c.cly0);
    if (!rc) rc = upload_a(h, clx1, &c.clx1);
    if (!rc) rc = upload_a(h, cly1, &c.cly1);
    // the tail sums of the wave kernel, one per tick index and centreline table (long double: they replace a 64-lane fp64 reduction)
    auto tail_table = [&](const std::vector<double>& cl, const double** dst) -> int {
        const int nt = c.ncl - p->P + 1;
        std::vector<double> T(std::max(nt, 1), 0.0);
        for (int j = 0; j < nt; ++j) {
            long double s = 0.0L;
            for (int i = p->C + 1; i <= p->P; ++i) s += (long double)wt[i - (p->C + 1)] * (long double)cl[j + i - 1];
            T[j] = (double)(s + (long double)c.wP * (long double)cl[p->P - 1]);
        }
        return upload_a(h, T, dst);
    };
    if (!rc) rc = tail_table(clx0, &c.tlx0);
    if (!rc) rc = tail_table(cly0, &c.tly0);
    if (!rc) rc = tail_table(clx1, &c.tlx1);
    if (!rc) rc = tail_table(cly1, &c.tly1);
    if (!rc) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) rc = fail_a(-2, "hipGetDeviceProperties failed");
        else {
            h->slots = prop.multiProcessorCount * (c.sinv_in_lds ? 1 : 4);   // persistent grid: workgroups per CU
            const int scratch_slots = prop.multiProcessorCount * 4;          // the slab always covers the 4-per-CU grid (the LDS variant may fall back to it)
            h->wave_blocks = prop.multiProcessorCount * 4; h->cus = prop.multiProcessorCount;
            if (hipMalloc((void**)&h->work_counter, 4 * sizeof(int)) != hipSuccess) rc = fail_a(-3, "counter allocation failed");
            else h->allocs.push_back(h->work_counter);
            if (!rc) { if (hipMalloc((void**)&h->c_dev, sizeof(DevA)) != hipSuccess) rc = fail_a(-3, "constants allocation failed"); else h->allocs.push_back(h->c_dev); }
            if (const char* e = std::getenv("ISMPC_A_KERNEL")) h->use_wave = std::strcmp(e, "block") != 0;
            void* sc = nullptr;
            if (hipMalloc(&sc, (size_t)scratch_slots * c.ldq * c.ldq * sizeof(double)) != hipSuccess) rc = fail_a(-3, "scratch allocation failed");
            else { h->allocs.push_back(sc); c.scratch = static_cast<double*>(sc); }
        }
    }
    if (!rc && c.sinv_in_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ismpc_a_tick_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)c.ldq * c.ldq * sizeof(double))) != hipSuccess) c.sinv_in_lds = 0, h->slots *= 4;
    }
    if (!rc && hipMemcpy(h->c_dev, &h->c, sizeof(DevA), hipMemcpyHostToDevice) != hipSuccess) rc = fail_a(-2, "constants upload failed");
    if (rc) { ismpc_a_destroy(h); return rc; }
    h->c_dirty = false;
    *out = h;
    return 0;
}

void ismpc_a_destroy(ismpc_a_handle* h)
{
    if (!h) return;
    DeviceGuardA guard_(h->device);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->prev) (void)hipFree(h->prev);
    if (h->hist) (void)hipFree(h->hist);
    if (h->defer_list) (void)hipFree(h->defer_list);
    if (h->order) (void)hipFree(h->order);
    if (h->pre) (void)hipFree(h->pre);
    if (h->feet_base) (void)hipFree(h->feet_base);
    delete h;
}

int ismpc_a_initial_state(const ismpc_a_handle* h, double disp_C, ismpc_a_state* st)
{
    if (!h || !st) return fail_a(-1, "null argument");
    std::memset(st, 0, sizeof(*st));
    st->x = disp_C / 2; st->xz = disp_C / 2;                  // :52-57
    st->cur_x = h->fsx[0]; st->cur_y = h->fsy[0];             // :58-59
    st->fc = 1; st->j = 1;
    return 0;
}

int ismpc_a_add_plan(ismpc_a_handle* h, const double* center)
{
    if (!h || !center) return fail_a(-1, "null argument");
    if (h->c.nplans >= 4) return fail_a(-1, "at most 4 base plans per handle");
    ON_DEVICE_A(h);
    std::vector<double> px(h->p.n_gait), py(h->p.n_gait);
    for (int i = 0; i < h->p.n_gait; ++i) { px[i] = center[i * 2]; py[i] = center[i * 2 + 1]; }
    const int k = h->c.nplans;
    int rc = upload_a(h, px, &h->c.plan_x[k]);
    if (!rc) rc = upload_a(h, py, &h->c.plan_y[k]);
    if (rc) return rc;
    h->c.nplans = k + 1;
    HIP_TRY_A(hipMemcpy(h->c_dev, &h->c, sizeof(DevA), hipMemcpyHostToDevice));
    return k;
}

int ismpc_a_reserve(ismpc_a_handle* h, int max_batch)
{
    if (!h || max_batch < 0) return fail_a(-1, "bad argument");
    ON_DEVICE_A(h);
    if (max_batch > h->prev_cap) {
        if (h->prev) HIP_TRY_A(hipFree(h->prev));
        h->prev = nullptr; h->prev_cap = 0;
        HIP_TRY_A(hipMalloc((void**)&h->prev, sizeof(ismpc_a_state) * (size_t)max_batch));
        h->prev_cap = max_batch;
    }
    if (max_batch > h->order_cap) {
        if (h->order) HIP_TRY_A(hipFree(h->order));
        h->order = nullptr; h->order_cap = 0;
        HIP_TRY_A(hipMalloc((void**)&h->order, sizeof(int) * (4 * (size_t)max_batch + 4)));
        h->order_cap = max_batch;
    }
    if (max_batch > h->pre_cap) {
        if (h->pre) HIP_TRY_A(hipFree(h->pre));
        h->pre = nullptr; h->pre_cap = 0;
        HIP_TRY_A(hipMalloc((void**)&h->pre, sizeof(ismpc_a::PiPre) * (size_t)max_batch));
        h->pre_cap = max_batch;
    }
    if (max_batch > h->defer_cap) {
        if (h->defer_list) HIP_TRY_A(hipFree(h->defer_list));
        h->defer_list = nullptr; h->defer_cap = 0;
        HIP_TRY_A(hipMalloc((void**)&h->defer_list, sizeof(int) * 2 * (size_t)max_batch));
        h->defer_cap = max_batch;
    }
    if (max_batch > h->hist_cap) {
        if (h->hist) HIP_TRY_A(hipFree(h->hist));
        h->hist = nullptr; h->hist_cap = 0; h->hist_valid = false;
        HIP_TRY_A(hipMalloc((void**)&h->hist, sizeof(unsigned long long) * 16 * (size_t)max_batch));
        h->hist_cap = max_batch;
    }
    return 0;
}

// The handle's scratch (prev, hist, order, defer_list) outlives the call that allocated it and is used by later calls on
// whatever stream those pass.  Before it is re-allocated on stream `s`, the previous launch's stream -- if it is another
// one -- is drained: the free can then not overtake kernels that still use the block.
static hipError_t grow_sync(ismpc_a_handle* h, hipStream_t s)
{
    if (h->used && h->last_stream != s) return hipStreamSynchronize(h->last_stream);
    return hipSuccess;
}

static int tick_launch(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev, const double* push_dev,
                       ismpc_a_out* out_dev, void* stream, int history = -1)
{
    if (!h || batch < 0 || (batch > 0 && !state_dev)) return fail_a(-1, "bad argument");
    if (batch == 0) return 0;
    ON_DEVICE_A(h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // history: -1 = as set by ismpc_a_set_warm_history, 0 = none, 1 = first tick of a rollout (store only), 2 = load + store
    if (history < 0) history = h->hist_ticks ? ((h->hist_valid && h->hist_batch == batch) ? 2 : 1) : 0;
    if (h->c.warm_add <= 0 || h->hist_off) history = 0;
    unsigned long long* hist = nullptr;
    if (history > 0) {
        if (batch > h->hist_cap) {                       // stream-ordered growth; ismpc_a_reserve sizes it beforehand
            HIP_TRY_A(grow_sync(h, s)); if (h->hist) HIP_TRY_A(hipFreeAsync(h->hist, s));
            h->hist = nullptr; h->hist_cap = 0; h->hist_valid = false;
            HIP_TRY_A(hipMallocAsync((void**)&h->hist, sizeof(unsigned long long) * 16 * (size_t)batch, s));
            h->hist_cap = batch;
        }
        hist = h->hist;
        if (history == 2 && !(h->hist_valid && h->hist_batch == batch)) history = 1;
        h->hist_valid = true; h->hist_batch = batch;
    }
    const int hist_load = history == 2 ? 1 : 0;
    struct Mark { ismpc_a_handle* h; hipStream_t s; ~Mark() { h->last_stream = s; h->used = true; } } mark_{h, s};
    if (batch > h->prev_cap) {
        HIP_TRY_A(grow_sync(h, s)); if (h->prev) HIP_TRY_A(hipFreeAsync(h->prev, s));
        h->prev = nullptr; h->prev_cap = 0;
        HIP_TRY_A(hipMallocAsync((void**)&h->prev, sizeof(ismpc_a_state) * (size_t)batch, s));
        h->prev_cap = batch;
    }
    if (inst_dev && batch > h->pre_cap) {
        HIP_TRY_A(grow_sync(h, s)); if (h->pre) HIP_TRY_A(hipFreeAsync(h->pre, s));
        h->pre = nullptr; h->pre_cap = 0;
        HIP_TRY_A(hipMallocAsync((void**)&h->pre, sizeof(ismpc_a::PiPre) * (size_t)batch, s));
        h->pre_cap = batch;
    }
    hipLaunchKernelGGL(ismpc_a_tick_prologue, dim3((batch + 255) / 256), dim3(256), 0, s, (const ismpc_a_state*)state_dev, h->prev, out_dev, batch,
                       (h->use_wave || inst_dev) ? h->work_counter : nullptr, inst_dev, inst_dev ? h->pre : nullptr, h->c.grav, h->c.dt, h->c.C, h->c.P);
    if (h->use_wave || inst_dev) {
        // structured solver, one wavefront per QP, 4 per workgroup; persistent grid (ismpc_a_wave.hpp)
        const int rl = (h->c.C + 63) / 64;
        if (h->c_dirty) { HIP_TRY_A(hipMemcpyAsync(h->c_dev, &h->c, sizeof(DevA), hipMemcpyHostToDevice, s)); h->c_dirty = false; }
        ismpc_a::WaveLaunch WL{h->c_dev, h->c.F, h->prev, state_dev, inst_dev, push_dev, out_dev, batch, h->work_counter, hist, hist_load,
                               h->precision, h->cus, h->wave_occ, nullptr, nullptr, 1, 0, 0, nullptr, nullptr, 0, s};
        // QPs per work-counter atomic (scripts/claim_sweep.sh): one device-wide atomic per QP costs 10-50 % when QPs are short (two rows
        // per lane, the fp32 solve at three, closed-loop ticks); pairs coarsen the balance too much when they are long
        WL.static_q = h->static_q; WL.pre = inst_dev ? h->pre : nullptr;
        WL.claim_chunk = h->claim_chunk > 0 ? h->claim_chunk : ((rl <= 2 || (h->precision == 1 && rl == 3) || hist_load) ? 2 : 1);
        hipError_t werr = hipSuccess;
        int wrc = -1;
        auto go = [&](const ismpc_a::WaveLaunch& W) {
            switch (rl) {
                case 1: case 2: return ismpc_a::launch_wave_rl2(W, &werr);
                case 3: return ismpc_a::launch_wave_rl3(W, &werr);
                case 4: return ismpc_a::launch_wave_rl4(W, &werr);
                default: return -1;
            }
        };
        if (inst_dev && h->c.F > 3 && h->bucket_by_F && rl <= 4 && h->c.F <= 6) {
            if (batch > h->order_cap) {
                HIP_TRY_A(grow_sync(h, s)); if (h->order) HIP_TRY_A(hipFreeAsync(h->order, s));
                h->order = nullptr; h->order_cap = 0;
                HIP_TRY_A(hipMallocAsync((void**)&h->order, sizeof(int) * (4 * (size_t)batch + 4), s));
                h->order_cap = batch;
            }
            int* counts = h->order + 4 * (size_t)h->order_cap;
            HIP_TRY_A(hipMemsetAsync(counts, 0, 4 * sizeof(int), s));
            hipLaunchKernelGGL(ismpc_a_bucket_by_F, dim3((batch + 255) / 256), dim3(256), 0, s, inst_dev, batch, h->c.F, h->order, h->order_cap, counts);
            wrc = 0;
            for (int f = 3; f <= h->c.F && wrc == 0; ++f) {
                if (f > 3) HIP_TRY_A(hipMemsetAsync(h->work_counter, 0, sizeof(int), s));
                ismpc_a::WaveLaunch W = WL;
                W.F = f; W.order = h->order + (size_t)(f - 3) * h->order_cap; W.count_ptr = counts + (f - 3);
                wrc = go(W);
            }
        } else {
            const bool resolve = h->precision == 1 && !h->defer_off;
            if (resolve) {
                if (batch > h->defer_cap) {                  // stream-ordered growth, as the history
                    HIP_TRY_A(grow_sync(h, s)); if (h->defer_list) HIP_TRY_A(hipFreeAsync(h->defer_list, s));
                    h->defer_list = nullptr; h->defer_cap = 0;
                    HIP_TRY_A(hipMallocAsync((void**)&h->defer_list, sizeof(int) * 2 * (size_t)batch, s));
                    h->defer_cap = batch;
                }
                WL.defer_list = h->defer_list; WL.defer_count = h->work_counter + 2;
            }
            wrc = go(WL);
            if (wrc == 0 && resolve) {
                // the QPs the fp32 launch handed over (block-solve check failed: a horizon pinned end to end), solved by the fp64
                // instantiation: usually nothing to do
                ismpc_a::WaveLaunch W2 = WL;
                W2.precision = 0; W2.work_counter = h->work_counter + 1; W2.order = h->defer_list; W2.count_ptr = h->work_counter + 2;
                W2.order_is_qp = 1; W2.defer_list = nullptr; W2.defer_count = nullptr; W2.static_q = 0; W2.claim_chunk = 1;
                // usually nothing to do (one QP in 30 000 on the bench pushes), but harder pushes hand over hundreds: up to 64 workgroups
                // (256 QPs at a time); a workgroup that finds the list empty exits after its prologue
                W2.grid_cap = h->resolve_grid;
                wrc = go(W2);
            }
        }
        if (wrc == 0) return 0;
        if (wrc == -2) return fail_a(-2, std::string("wave kernel launch: ") + hipGetErrorString(werr));
        if (inst_dev) return fail_a(-1, "per-instance gait parameters need the structured kernel: 3 <= F <= 6 and C <= 256");
        if (h->precision != 0) return fail_a(-1, "the fp32 solve needs the structured kernel: 3 <= F <= 6 and C <= 256");
    }
    const int grid = std::min(2 * batch, h->slots);
    hipLaunchKernelGGL(ismpc_a_tick_kernel, dim3(grid), dim3(T), h->c.sinv_in_lds ? (size_t)h->c.ldq * h->c.ldq * sizeof(double) : 0, s, h->c, (const ismpc_a_state*)h->prev, state_dev, push_dev, out_dev, batch);
    HIP_TRY_A(hipGetLastError());
    return 0;
}

int ismpc_a_tick_batch_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const double* push_dev,
                              ismpc_a_out* out_dev, void* stream)
{
    return tick_launch(h, batch, state_dev, nullptr, push_dev, out_dev, stream);
}

int ismpc_a_set_precision(ismpc_a_handle* h, int fp32)
{
    if (!h) return fail_a(-1, "null handle");
    if (fp32 && (h->c.F < 3 || h->c.F > 6 || !h->use_wave)) return fail_a(-1, "the fp32 solve needs the structured kernel: 3 <= F <= 6");
    h->precision = fp32 ? 1 : 0; h->hist_valid = false;
    return 0;
}

int ismpc_a_last_deferred(ismpc_a_handle* h)
{
    if (!h) return fail_a(-1, "null handle");
    if (!h->used || h->precision != 1 || h->defer_off) return 0;
    ON_DEVICE_A(h);
    int n = 0;
    HIP_TRY_A(hipStreamSynchronize(h->last_stream));
    HIP_TRY_A(hipMemcpy(&n, h->work_counter + 2, sizeof(int), hipMemcpyDeviceToHost));
    return n;
}

int ismpc_a_set_warm_history(ismpc_a_handle* h, int enabled)
{
    if (!h) return fail_a(-1, "null handle");
    h->hist_ticks = enabled != 0; h->hist_valid = false;
    return 0;
}

int ismpc_a_tick_batch_inst_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev,
                                   const double* push_dev, ismpc_a_out* out_dev, void* stream)
{
    if (batch > 0 && !inst_dev) return fail_a(-1, "null per-instance parameter array");
    return tick_launch(h, batch, state_dev, inst_dev, push_dev, out_dev, stream);
}

int ismpc_a_rollout_inst_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev, int ticks,
                                ismpc_a_out* out_traj_dev, void* stream)
{
    if (!h || batch < 0 || ticks < 0) return fail_a(-1, "bad argument");
    for (int t = 0; t < ticks; ++t) {
        if (batch > 0 && !inst_dev) return fail_a(-1, "null per-instance parameter array");
        int rc = tick_launch(h, batch, state_dev, inst_dev, nullptr, out_traj_dev ? out_traj_dev + (size_t)t * batch : nullptr, stream, t == 0 ? 1 : 2);
        if (rc) return rc;
    }
    return 0;
}


int ismpc_a_feet_rows(const ismpc_a_handle* h) { return h ? h->feet.rows : -1; }

int ismpc_a_feet_init_device(ismpc_a_handle* h, const ismpc_a_gait* g, const double* foot_plan_host, int rows, int batch,
                             double* feet_dev, void* stream)
{
    if (!h || !g || !foot_plan_host || rows < 2 || batch < 0 || (batch > 0 && !feet_dev)) return fail_a(-1, "bad argument");
    ON_DEVICE_A(h);
    const int rp = rows + 8;                                       // the walk script writes rows fc+1 .. fc+8
    std::vector<double> base((size_t)rp * 8);
    for (int r = 0; r < rp; ++r) std::memcpy(&base[(size_t)r * 8], foot_plan_host + (size_t)std::min(r, rows - 1) * 8, 64);
    if (h->feet_base) { (void)hipFree(h->feet_base); h->feet_base = nullptr; }
    HIP_TRY_A(hipMalloc((void**)&h->feet_base, base.size() * sizeof(double)));
    HIP_TRY_A(hipMemcpy(h->feet_base, base.data(), base.size() * sizeof(double), hipMemcpyHostToDevice));
    h->feet.gait = g->gait; h->feet.rows = rp; h->feet.phi = g->phi; h->feet.disp_i = g->disp_i; h->feet.disp_o = g->disp_o; h->feet.disp_forw = g->disp_forw;
    if (batch > 0) {
        hipLaunchKernelGGL(ismpc_a_feet_fill, dim3(std::min(1024, (batch * rp * 8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           (const double*)h->feet_base, feet_dev, rp, batch);
        HIP_TRY_A(hipGetLastError());
    }
    return 0;
}

static int tick_feet(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev, const double* push_dev,
                     ismpc_a_out* out_dev, double* feet_dev, void* stream, int history = -1)
{
    if (!h || !out_dev || (batch > 0 && !feet_dev) || h->feet.rows == 0) return fail_a(-1, "feet: call ismpc_a_feet_init_device first and pass an output buffer");
    if (inst_dev && h->feet_plans == 0) return fail_a(-1, "feet: per-instance batches need ismpc_a_feet_init_inst_device");
    ON_DEVICE_A(h);                                                 // the feet launch below runs on the handle's device too
    int rc = tick_launch(h, batch, state_dev, inst_dev, push_dev, out_dev, stream, history);
    if (rc || batch == 0) return rc;
    FeetParamsSet fs = h->feet_set;
    if (!inst_dev) fs.p[0] = h->feet;
    hipLaunchKernelGGL(ismpc_a_feet_kernel, dim3((batch + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       fs, inst_dev, inst_dev ? h->feet_plans : 1, (const ismpc_a_state*)h->prev, (const ismpc_a_out*)out_dev, feet_dev, batch);
    HIP_TRY_A(hipGetLastError());
    return 0;
}

int ismpc_a_tick_feet_batch_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const double* push_dev,
                                   ismpc_a_out* out_dev, double* feet_dev, void* stream)
{
    return tick_feet(h, batch, state_dev, nullptr, push_dev, out_dev, feet_dev, stream);
}

int ismpc_a_rollout_feet_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, int ticks, ismpc_a_out* out_traj_dev,
                                double* feet_dev, void* stream)
{
    if (!h || !out_traj_dev || batch < 0 || ticks < 0) return fail_a(-1, "bad argument");
    int rc = 0;                                                      // closed loop: previous working set as the first guess
    for (int t = 0; t < ticks && !rc; ++t)
        rc = tick_feet(h, batch, state_dev, nullptr, nullptr, out_traj_dev + (size_t)t * batch, feet_dev, stream, t == 0 ? 1 : 2);
    return rc;
}

// Per-instance gait parameters (ismpc_a_inst): instance b follows the foot rules and starts from the foot plan of its base plan.
int ismpc_a_feet_init_inst_device(ismpc_a_handle* h, const ismpc_a_gait* gaits, const double* foot_plans_host, int rows, int nplans,
                                  int batch, const ismpc_a_inst* inst_dev, double* feet_dev, void* stream)
{
    if (!h || !gaits || !foot_plans_host || rows < 2 || nplans < 1 || nplans > 4 || batch < 0 || (batch > 0 && (!feet_dev || !inst_dev)))
        return fail_a(-1, "bad argument");
    if (nplans != h->c.nplans) return fail_a(-1, "feet: one gait record and one foot plan per base plan of the handle (ismpc_a_create + ismpc_a_add_plan)");
    ON_DEVICE_A(h);
    const int rp = rows + 8;                                       // the walk script writes rows fc+1 .. fc+8
    std::vector<double> base((size_t)nplans * rp * 8);
    for (int k = 0; k < nplans; ++k)
        for (int r = 0; r < rp; ++r)
            std::memcpy(&base[((size_t)k * rp + r) * 8], foot_plans_host + ((size_t)k * rows + std::min(r, rows - 1)) * 8, 64);
    if (h->feet_base) { (void)hipFree(h->feet_base); h->feet_base = nullptr; }
    HIP_TRY_A(hipMalloc((void**)&h->feet_base, base.size() * sizeof(double)));
    HIP_TRY_A(hipMemcpy(h->feet_base, base.data(), base.size() * sizeof(double), hipMemcpyHostToDevice));
    for (int k = 0; k < nplans; ++k) {
        FeetParams& f = h->feet_set.p[k];
        f.gait = gaits[k].gait; f.rows = rp; f.phi = gaits[k].phi; f.disp_i = gaits[k].disp_i; f.disp_o = gaits[k].disp_o; f.disp_forw = gaits[k].disp_forw;
    }
    h->feet_plans = nplans; h->feet = h->feet_set.p[0];
    if (batch > 0) {
        hipLaunchKernelGGL(ismpc_a_feet_fill_inst, dim3(std::min(1024, (int)(((size_t)batch * rp * 8 + 255) / 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
                           (const double*)h->feet_base, inst_dev, nplans, feet_dev, rp, batch);
        HIP_TRY_A(hipGetLastError());
    }
    return 0;
}

int ismpc_a_tick_feet_batch_inst_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev, const double* push_dev,
                                        ismpc_a_out* out_dev, double* feet_dev, void* stream)
{
    if (batch > 0 && !inst_dev) return fail_a(-1, "null per-instance parameter array");
    return tick_feet(h, batch, state_dev, inst_dev, push_dev, out_dev, feet_dev, stream);
}

int ismpc_a_rollout_feet_inst_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const ismpc_a_inst* inst_dev, int ticks,
                                     ismpc_a_out* out_traj_dev, double* feet_dev, void* stream)
{
    if (!h || !out_traj_dev || batch < 0 || ticks < 0 || (batch > 0 && !inst_dev)) return fail_a(-1, "bad argument");
    int rc = 0;
    for (int t = 0; t < ticks && !rc; ++t)
        rc = tick_feet(h, batch, state_dev, inst_dev, nullptr, out_traj_dev + (size_t)t * batch, feet_dev, stream, t == 0 ? 1 : 2);
    return rc;
}

// quad_as_bip_no_plots.m:482-509 / quad_walk_no_plots.m:562-613 (host)
int ismpc_a_foot_trajectories(const ismpc_a_gait* g, int step, const double* foot_plan, int rows, int sim_duration, double* dst)
{
    if (!g || !foot_plan || !dst || step < 1 || sim_duration < step) return fail_a(-1, "bad argument");
    const int nsteps = sim_duration / step, n = nsteps * step;
    if (nsteps + 1 > rows) return fail_a(-1, "foot_plan has too few rows for this duration");
    auto FPL = [&](int r, int c) { return foot_plan[(size_t)(r - 1) * 8 + (c - 1)]; };
    auto put = [&](int foot, int row, double x, double y, double z) { double* d = dst + ((size_t)foot * n + row) * 3; d[0] = x; d[1] = y; d[2] = z; };
    int row = 0, cont = 1;
    for (int i = 1; i <= nsteps; ++i) {
        if (g->gait == 0) {
            if (step <= 50) return fail_a(-1, "the trot writer assumes step_duration > 50 (30 + 50 rows per step)");
            for (int k = 1; k <= step - 50; ++k, ++row) {
                put(0, row, FPL(i,7), FPL(i,8), 0.0); put(3, row, FPL(i,3), FPL(i,4), 0.0); put(1, row, FPL(i,5), FPL(i,6), 0.0); put(2, row, FPL(i,1), FPL(i,2), 0.0);
            }
            for (int j = 1; j <= 50; ++j, ++row) {
                const double z = -0.000032 * j * j + 0.0016 * j;
                const int still1 = (i % 2 == 1) ? 7 : 1, still2 = (i % 2 == 1) ? 3 : 5, mv1 = (i % 2 == 1) ? 1 : 7, mv2 = (i % 2 == 1) ? 5 : 3;
                const int footOf[9] = {0, 2, 0, 3, 0, 1, 0, 0, 0};             // column -> file index (fl 0, fr 1, rl 2, rr 3)
                put(footOf[still1], row, FPL(i,still1), FPL(i,still1+1), 0.0); put(footOf[still2], row, FPL(i,still2), FPL(i,still2+1), 0.0);
                put(footOf[mv1], row, FPL(i,mv1) + (FPL(i+1,mv1) - FPL(i,mv1)) / 50 * j, FPL(i,mv1+1) + (FPL(i+1,mv1+1) - FPL(i,mv1+1)) / 50 * j, z);
                put(footOf[mv2], row, FPL(i,mv2) + (FPL(i+1,mv2) - FPL(i,mv2)) / 50 * j, FPL(i,mv2+1) + (FPL(i+1,mv2+1) - FPL(i,mv2+1)) / 50 * j, z);
            }
        } else {
            for (int k = 1; k <= step; ++k, ++row) {
                const double z = -0.000032 * k * k + 0.0016 * k;
                const int mv = (cont == 2) ? 7 : (cont == 4) ? 3 : (cont == 6) ? 5 : (cont == 8) ? 1 : 0;
                const int cols[4] = {7, 5, 1, 3};
                for (int ft = 0; ft < 4; ++ft) {
                    const int cc = cols[ft];
                    if (cc == mv) put(ft, row, FPL(i,cc) + (FPL(i+1,cc) - FPL(i,cc)) / step * k, FPL(i,cc+1) + (FPL(i+1,cc+1) - FPL(i,cc+1)) / step * k, z);
                    else put(ft, row, FPL(i,cc), FPL(i,cc+1), 0.0);
                }
            }
            cont = (cont == 8) ? 1 : cont + 1;
        }
    }
    return n;
}

// fprintf(file, '%d %d %d\n', row): MATLAB prints an integer-valued double with %d and anything else with %e
int ismpc_a_write_trajectory_txt(const char* path, const double* rows3, int n)
{
    if (!path || !rows3 || n < 0) return fail_a(-1, "bad argument");
    FILE* f = std::fopen(path, "w");
    if (!f) return fail_a(-1, std::string("cannot open ") + path);
    for (int r = 0; r < n; ++r) {
        for (int c = 0; c < 3; ++c) {
            const double v = rows3[(size_t)r * 3 + c];
            if (v == std::floor(v) && std::fabs(v) < 1e15) std::fprintf(f, "%lld", (long long)v); else std::fprintf(f, "%e", v);
            std::fputc(c == 2 ? '\n' : ' ', f);
        }
    }
    std::fclose(f);
    return 0;
}

int ismpc_a_rollout_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, int ticks, ismpc_a_out* out_traj_dev, void* stream)
{
    if (!h || batch < 0 || ticks < 0 || (batch > 0 && !state_dev)) return fail_a(-1, "bad argument");
    for (int t = 0; t < ticks; ++t) {
        int rc = tick_launch(h, batch, state_dev, nullptr, nullptr, out_traj_dev ? out_traj_dev + (size_t)t * batch : nullptr, stream, t == 0 ? 1 : 2);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
