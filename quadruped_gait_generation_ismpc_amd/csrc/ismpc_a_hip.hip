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
//  * ismpc_a_tick_wave<RL, F, PI> (default): ONE WAVEFRONT per QP, nothing of working-set size is stored.  The Gram
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
#include "../../include/ismpc_a.h"

namespace {

constexpr int T = 256;                 // threads per workgroup; requires C + F <= 256
constexpr int MAXF = 8;
constexpr int QCAP = 264;              // capacity of the working set (>= C + F + 1)

struct DevA {
    int C, P, F, step, ds, n_gait, ncl, ldq, max_iter, sinv_in_lds;
    int warm_add, warm_drop, warm_extra; // block warm start of the wave kernel: passes that add + drop rows, passes that only drop, re-entries
    double dt, eta, w, Qf, disp_forw, disp_forw_dummy, disp_L, aa, wP, sumw;
    double Au[9], Bu[3];
    const double *a, *PA, *wtail;      // stability row, its prefix sums PA[i] = sum_{k<i} a_k, tail weights (index i-(C+1))
    const double *fsx, *fsy;           // base plan, 0-based (fs_plan(k+1))
    const double *clx0, *cly0, *clx1, *cly1;   // centreline: initial / rebuilt structure, 0-based (cl(k+1))
    double* scratch;                   // per-workgroup S^-1 : ldq x ldq doubles each
    const double *plan_x[4], *plan_y[4]; int nplans; double grav;   // base plans selectable per instance (ismpc_a_inst.plan)
};

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

// =====================================================================================================
// Wavefront-per-QP kernel: the STRUCTURED dual active-set solver (scripts/proto_structured.py is its numpy model).
//
// No matrix over the working set exists.  For the active ZMP rows, sorted by sample index i_1 < i_2 < ..., the
// Gram block in the H^-1 metric is dt^2 min(i_j, i_k) + (footstep coupling)/Qf.  dt^2 min(.,.) is the covariance of
// a random walk, so its inverse is TRIDIAGONAL: (K^-1 y)_j = (y_j - y_prev)/g_j - (y_next - y_j)/g_next with g the
// index gaps -- each active row only needs its previous / next active row -- and K^-1 applied to a kernel column
// min(i+, .) is linear interpolation at i+ (two non-zeros).  Everything else -- the footstep coupling M~ (F columns),
// the stability row and the F kinematic rows -- is a rank <= 2F+1 border: V_i = [M~_i, dt PA_i, Bk_i] has a closed
// form per row, G = V' K^-1 V / dt^2 (m x m, m = 2F+1) is kept by +/- one outer product per gap created/destroyed,
// and one quasi-definite m x m system per step gives the border unknowns.  Step lengths, add / drop logic and
// termination are Goldfarb-Idnani's.  Lanes own RL consecutive rows (row = ZMP sample), so the primal direction is
// "own multiplier as an impulse + one suffix scan" with no scatter; wave collectives are DPP scans / min / max.
// =====================================================================================================
// Per-wavefront LDS.  Small vectors (length m = 2F+1 or F+2) are kept ONE ELEMENT PER LANE in registers and mirrored
// here when other lanes need them by index; nothing of size "working set" is stored anywhere.
template <int F> struct WaveLds {
    static constexpr int m = 2 * F + 1;
    double sv[T];                      // V_row . y of every ZMP row
    double w1s[T];                     // mapping weight of every row (wave-uniform look-ups by row index)
    int    k1s[T];                     // first mapped footstep of every row
    double comb[F + 2];                // footstep-column coefficients seen by a row: comb[k1], comb[k1+1]
    double fl[F + 2];                  // f[0..F+1] with fl[0] = fl[F+1] = 0
    double pf[F + 2];                  // the plan's footsteps (same layout): block warm start
    double th[F * (F + 1) / 2 + 2 * F + 2];   // Gram sums of the block warm start: Theta (upper triangle), psi, gamma, sigma, gamma_E
    double G[m * m];                   // V' K^-1 V / dt^2 over the active ZMP rows
    double vp[m], hx[m], d1[m], d2[m], d0[m], cc[m], mt[m];
};
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <int CTRL, int RM> __device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xf, false); }
__device__ __forceinline__ int wave_scan_max_i(int v)      // inclusive prefix maximum over the 64 lanes
{
    const int lo = -2147483647 - 1;
    v = max(v, dpp_i<0x111, 0xf>(lo, v)); v = max(v, dpp_i<0x112, 0xf>(lo, v)); v = max(v, dpp_i<0x114, 0xf>(lo, v));
    v = max(v, dpp_i<0x118, 0xf>(lo, v)); v = max(v, dpp_i<0x142, 0xa>(lo, v)); v = max(v, dpp_i<0x143, 0xc>(lo, v));
    return v;
}
__device__ __forceinline__ double wave_min_d(double v)
{
    v = fmin(v, dpp64<0x111, 0xf, false>(INFINITY, v)); v = fmin(v, dpp64<0x112, 0xf, false>(INFINITY, v));
    v = fmin(v, dpp64<0x114, 0xf, false>(INFINITY, v)); v = fmin(v, dpp64<0x118, 0xf, false>(INFINITY, v));
    v = fmin(v, dpp64<0x142, 0xa, false>(INFINITY, v)); v = fmin(v, dpp64<0x143, 0xc, false>(INFINITY, v));
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v)
{
    v = wave_scan_up(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double rl_d(double v, int l)
{
    const int ll = __builtin_amdgcn_readfirstlane(l);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), ll), __builtin_amdgcn_readlane(__double2loint(v), ll));
}
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }

// 1/x to rounding error: v_rcp_f64 + two Newton steps (the IEEE division sequence is more than twice as long, and the
// active-set loop divides by gaps and pivots on its critical path)
__device__ __forceinline__ double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// value held by the owner of ZMP row `row` (1-based, wave-uniform) in a per-row register array
template <int RL> __device__ __forceinline__ double at_row(const double (&v)[RL], int row)
{
    const int o = (row - 1) / RL, k = (row - 1) - o * RL;
    double x = v[0];
#pragma unroll
    for (int r = 1; r < RL; ++r) if (k == r) x = v[r];
    return rl_d(x, o);
}
template <int RL> __device__ __forceinline__ int at_row(const int (&v)[RL], int row)
{
    const int o = (row - 1) / RL, k = (row - 1) - o * RL;
    int x = v[0];
#pragma unroll
    for (int r = 1; r < RL; ++r) if (k == r) x = v[r];
    return rl_i(x, o);
}
// element `e` (this lane's) of the border row V = [M~ (F), dt PA, Bk (F)] of a ZMP row with mapping (k1, w1, 1-w1), PA = pa
template <int F> __device__ __forceinline__ double border_elem(int e, int k1, double w1, double pa, double dt, double isq)
{
    if (e == F) return dt * pa;
    const double w2 = 1.0 - w1;
    const int r = e < F ? e + 1 : e - F;                  // footstep column 1..F
    const double mr = (r == k1) ? w1 : ((r == k1 + 1) ? w2 : 0.0);
    if (e < F) return mr * isq;
    const double mp = (r - 1 >= 1) ? ((r - 1 == k1) ? w1 : ((r - 1 == k1 + 1) ? w2 : 0.0)) : 0.0;
    return (-mr + mp) * isq;
}

// centreline value cl(k0+1) without the table: quad_walk_no_plots.m:86-99 (initial structure) / :540-549 (rebuilt one),
// linspace as MATLAB evaluates it.  Used when step / ds differ per instance.
__device__ __forceinline__ double cl_closed(const double* __restrict__ fs, int step, int ds, bool rebuilt, int k0)
{
    const int s = k0 / step, r = k0 - s * step, q = r - (step - ds);
    const double d1 = fs[s];
    if (q <= 0 || (rebuilt && s == 0)) return d1;
    const double d2 = fs[s + 1];
    if (q == ds - 1) return d2;
    return d1 + ((double)q * (d2 - d1)) / (double)(ds - 1);
}

// -DISMPC_A_PROF: per-phase shader-clock totals of the Goldfarb-Idnani loop (development aid, scripts/prof_a.py)
#ifdef ISMPC_A_PROF
__device__ unsigned long long g_prof[32];
#define PROF_T0() unsigned long long pt_ = __builtin_readcyclecounter()
#define PROF(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (lane == 0 && (work & 127) == 0) { atomicAdd(&g_prof[k_], n_ - pt_); atomicAdd(&g_prof[16 + (k_)], 1ull); } pt_ = __builtin_readcyclecounter(); } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF(k_) do {} while (0)
#endif

// RL = ZMP rows per lane (C <= 64 RL), F = footsteps in the horizon (m = 2F+1 border columns).
// PI: per-instance gait parameters (ismpc_a_inst): height, Qf, step, ds, F <= the template F, base plan.
// Residency target (workgroups per CU), measured on MI355X at batch 16 384 (scripts/occ_sweep.sh): the 128-VGPR budget of
// 4 per CU spills 80-220 registers per lane; 3 per CU is best for 2 rows per lane (walk C=100: 5.0e6 ticks/s vs 4.0e6),
// 2 per CU for 3-4 rows per lane (walk C=150: 2.2e6 vs 1.4e6; Monte-Carlo C=200: 1.17e6 vs 0.75e6).
#ifndef ISMPC_A_WAVE_MINBLOCKS
#define ISMPC_A_WAVE_MINBLOCKS (RL <= 2 ? 3 : 2)
#endif
template <int RL, int F, bool PI>
__global__ __launch_bounds__(T, ISMPC_A_WAVE_MINBLOCKS)
void ismpc_a_tick_wave(const DevA c, const ismpc_a_state* __restrict__ state_in, ismpc_a_state* __restrict__ state,
                       const ismpc_a_inst* __restrict__ ipar, const double* __restrict__ push, ismpc_a_out* __restrict__ out, int batch,
                       int* __restrict__ work_counter, unsigned long long* __restrict__ hist, int hist_load)
{
    constexpr int m = 2 * F + 1;
    __shared__ WaveLds<F> lds_all[T / 64];
    __shared__ double a_s[T], pa_s[T + 1];                  // stability row and its prefix sums: same for every QP of the handle
    __shared__ double a_pi[PI ? T / 64 : 1][PI ? T : 1], pa_pi[PI ? T / 64 : 1][PI ? T + 1 : 1];   // ... or one per wavefront
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds<F>& L = lds_all[wv];
    const int C = c.C, P = c.P;
    const double dt = c.dt, idt2 = 1.0 / (dt * dt);
    const bool klane = lane >= 1 && lane <= F;            // lane r owns kinematic row r (and f_r)
    if (!PI) {
        for (int k = threadIdx.x; k <= C; k += T) { pa_s[k] = c.PA[k]; if (k < C) a_s[k] = c.a[k]; }
        __syncthreads();
    }
    const double* ap = PI ? a_pi[PI ? wv : 0] : a_s;
    const double* pap = PI ? pa_pi[PI ? wv : 0] : pa_s;

    // QPs are claimed one at a time from a global counter: iteration counts differ a lot between instances
    for (;;) {
        int claimed = 0;
        if (lane == 0) claimed = atomicAdd(work_counter, 1);
        const int work = __builtin_amdgcn_readfirstlane(claimed);
        if (work >= 2 * batch) break;
        const int inst = work >> 1, axis = work & 1;
#ifdef ISMPC_A_PROF
        unsigned long long pq_ = __builtin_readcyclecounter();
#endif
        const ismpc_a_state st = state_in[inst];
        const double pos = axis == 0 ? st.x : st.y;
        const double vel = (axis == 0 ? st.xd : st.yd) + (push ? push[inst * 2 + axis] : 0.0);
        const double zmp = axis == 0 ? st.xz : st.yz;
        const double cur = axis == 0 ? st.cur_x : st.cur_y;
        const double off = axis == 0 ? st.off_x : st.off_y;
        const int j = st.j, fc = st.fc;
        int status = 0;
        // ---- gait parameters: the handle's, or this instance's
        int step_ = c.step, ds_ = c.ds, Fi = F, plan = 0;
        double Qf = c.Qf, eta = c.eta, aa = c.aa;
        if (PI) {
            const ismpc_a_inst ip = ipar[inst];
            step_ = ip.step; ds_ = ip.ds; Fi = ip.F; plan = ip.plan; Qf = ip.Qf;
            if (step_ < 2 || ds_ < 2 || ds_ >= step_ || Fi < 1 || Fi > F || plan < 0 || plan >= c.nplans || !(ip.height > 0) || !(Qf > 0)) {
                status |= ISMPC_A_ST_BAD_INDEX; step_ = 2; ds_ = 1; Fi = 1; plan = 0; Qf = 1.0; eta = 1.0;
            } else eta = sqrt(c.grav / ip.height);
        }
        const double sq = sqrt(Qf), isq = 1.0 / sq, iQf = 1.0 / Qf;
        const float rstep = 1.0f / (float)step_;
        const double* fs = PI ? (axis == 0 ? c.plan_x[plan] : c.plan_y[plan]) : (axis == 0 ? c.fsx : c.fsy);
        const double* cl = st.rebuilt ? (axis == 0 ? c.clx1 : c.cly1) : (axis == 0 ? c.clx0 : c.cly0);
        const double cloff = st.rebuilt ? off : 0.0;
        const int ncl = PI ? (c.n_gait - 1) * step_ : c.ncl;
        if (fc < 1 || fc + Fi > c.n_gait || j < 1 || j + P > ncl || j < step_ * (fc - 1) || j > step_ * fc - 1)
            status |= ISMPC_A_ST_BAD_INDEX;
        const double zlo0 = -(-1.0 * (-zmp - c.w / 2)), zhi0 = 1.0 * (-zmp + c.w / 2);       // band without the current-footstep term
        if (PI) {
            // stability row a_i (quad_walk_no_plots.m:233-238), its prefix sums and a'a for this instance's eta
            double* aw = a_pi[PI ? wv : 0]; double* paw = pa_pi[PI ? wv : 0];
            const double lam = exp(-eta * dt);
            const double k1c = (1 / eta) * (1 - lam) / (1 - pow(lam, (double)C)), k2c = dt * 1.0 * exp(-eta * dt * C);
            double av[RL], cum[RL], loc = 0.0, sqs = 0.0;
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const int i0 = lane * RL + k;
                av[k] = (i0 < C) ? k1c * exp(-eta * dt * i0) - k2c : 0.0;
                loc += av[k]; cum[k] = loc; sqs += av[k] * av[k];
            }
            const double base = wave_scan_up(loc) - loc;
#pragma unroll
            for (int k = 0; k < RL; ++k) { const int i0 = lane * RL + k; if (i0 < C) { aw[i0] = av[k]; paw[i0 + 1] = base + cum[k]; } }
            if (lane == 0) paw[0] = 0.0;
            aa = wave_sum_d(sqs);
            WAVE_LDS_SYNC();
        }

        // ---- per-row data: lane owns ZMP rows lane*RL+1 .. lane*RL+RL (row i = sample i, u index i-1)
        double u[RL], zlo[RL], zhi[RL], w1[RL], inrm[RL], mu[RL];
        int k1[RL], sta[RL], prv[RL], nxt[RL];
        bool ovf = false;
#pragma unroll
        for (int k = 0; k < RL; ++k) {
            const int i = lane * RL + k + 1;
            u[k] = 0.0; mu[k] = 0.0; sta[k] = 0; prv[k] = 0; nxt[k] = 0;
            if (i <= C) {
                int qd = (int)((float)(j + i) * rstep);                          // (j + i) / step_ without the integer-division sequence
                if (qd * step_ > j + i) --qd; else if ((qd + 1) * step_ <= j + i) ++qd;
                int pf = qd - fc + 1; if (pf < 0) pf = 0;
                const int rem = step_ * (fc + pf) - (j + i);
                w1[k] = (rem > ds_) ? 1.0 : (double)rem / ds_;                    // mapping(i, pf+1); the next column gets 1 - w1
                k1[k] = pf;
                ovf = ovf || pf > Fi || (rem <= ds_ && pf + 1 > Fi);
                const double m1 = (pf == 0) ? w1[k] : 0.0;
                zhi[k] = zhi0 + m1 * cur; zlo[k] = zlo0 + m1 * cur;
                const double w2 = 1.0 - w1[k];
                double mm = w2 * w2;                                              // |M_i|^2 over the footstep columns
                if (pf >= 1) mm += w1[k] * w1[k];
                { const double xn = dt * dt * (double)i + mm * iQf; double r_ = __builtin_amdgcn_rsq(xn); inrm[k] = r_ * (1.5 - 0.5 * xn * r_ * r_); }   // only ranks candidates
                L.k1s[i - 1] = pf; L.w1s[i - 1] = w1[k];
            } else { w1[k] = 1.0; k1[k] = 0; zlo[k] = -INFINITY; zhi[k] = INFINITY; inrm[k] = 0.0; }
        }
        if (__builtin_amdgcn_ballot_w64(ovf) != 0) status |= ISMPC_A_ST_OVERFLOW;
        // anticipative tail (quad_walk_no_plots.m:227-231)
        double tl = 0.0;
        if (!(status & ISMPC_A_ST_BAD_INDEX)) {
            if (PI) {
                const double om = 1 - exp(-eta * dt);
                for (int i = C + 1 + lane; i <= P; i += 64)
                    tl += exp(-eta * dt * i) * om * ((cl_closed(fs, step_, ds_, st.rebuilt != 0, j + i - 1) + cloff) - cur);
            } else
                for (int i = C + 1 + lane; i <= P; i += 64) tl += c.wtail[i - (C + 1)] * ((cl[j + i - 1] + cloff) - cur);
        }
        double tail = wave_sum_d(tl);
        if (!(status & ISMPC_A_ST_BAD_INDEX))
            tail += PI ? exp(-eta * dt * P) * ((cl_closed(fs, step_, ds_, st.rebuilt != 0, P - 1) + cloff) - cur)
                       : c.wP * ((cl[P - 1] + cloff) - cur);
        const double beq = pos + vel / eta - zmp - tail;
        // ---- kinematic row r and footstep f_r live in lane r (1..F); Khat_r = sqrt(Qf) (f_r - f_{r-1})
        double fr = 0.0, klo = -INFINITY, khi = INFINITY, muK = 0.0;
        int kact = 0;
        if (klane) {
            const int r = lane;
            double bup = axis == 0 ? c.disp_forw : (c.disp_L / 2 + c.disp_L / 2);
            if (fc == 1 && r == 1) bup = axis == 0 ? c.disp_forw_dummy : (c.disp_L / 2 + c.disp_L / 2);
            double blo = bup;
            if (r == 1) { bup = bup + cur; blo = blo - cur; }
            khi = bup; klo = -blo;
            fr = (status & ISMPC_A_ST_BAD_INDEX) ? 0.0 : fs[fc + r - 1] + off;
            if (PI && r > Fi) { khi = INFINITY; klo = -INFINITY; fr = 0.0; }      // beyond this instance's horizon: no variable, no row
        }
        const double knrm = (lane >= 2) ? sq * 0.70710678118654752440 : sq;   // 1 / |K_r|_{H^-1}: |kvec_r|^2 = 2 (r >= 2) or 1
        int iters = 0, qz = 0, qk = 0;
        double muE = 0.0;
        bool done_opt = false;                                // the block passes ended on a checked optimum: nothing left to do
#ifdef ISMPC_A_PROF
        { const unsigned long long n_ = __builtin_readcyclecounter(); if (lane == 0 && (work & 127) == 0) { atomicAdd(&g_prof[10], n_ - pq_); atomicAdd(&g_prof[26], 1ull); } pq_ = n_; }
#endif
        if (status == 0) {
            // ---- equality first: u = (b / a'a) a
            const double t0 = beq / aa;
#pragma unroll
            for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; u[k] = (i <= C) ? t0 * ap[i - 1] : 0.0; }
            muE = t0;
            for (int e = lane; e < m * m; e += 64) L.G[e] = 0.0;
            if (lane <= F + 1) L.pf[lane] = fr;
            WAVE_LDS_SYNC();

            // ---- small quasi-definite system  [[I+G11, G1x],[Gx1, Gxx - Sxx]] cc = rhs (L.hx), G in L.G; unknown order:
            // 0..F-1 footstep columns, F = stability row, F+1..2F = Khat_1..F (rows outside kmask: pinned to 0).  Returns
            // this lane's cc[lane] and leaves cc in L.cc.
            auto solve_small = [&](const unsigned long long kmask) __attribute__((always_inline)) -> double {
                // lane i < m owns row i of the augmented matrix in registers; the pivot row travels by readlane: no LDS
                // traffic and no barriers inside the elimination
                const int i = lane < m ? lane : m - 1;
                const bool ipin = i > F && !((kmask >> (i - F)) & 1ull);
                double Tr[m + 1];
#pragma unroll
                for (int jj = 0; jj < m; ++jj) {
                    double val = L.G[i * m + jj];
                    if (jj < F && i == jj) val += 1.0;
                    if (jj == F && i == F) val -= aa;
                    if (jj > F && i > F) {
                        const int r1 = i - F, r2 = jj - F;
                        val -= (r1 == r2) ? (r1 >= 2 ? 2.0 : 1.0) : ((r1 - r2 == 1 || r2 - r1 == 1) ? -1.0 : 0.0);
                    }
                    const bool jpin = jj > F && !((kmask >> (jj - F)) & 1ull);
                    if (ipin || jpin) val = (i == jj) ? -1.0 : 0.0;
                    Tr[jj] = val;
                }
                Tr[m] = ipin ? 0.0 : L.hx[i];
#pragma unroll
                for (int kk = 0; kk < m; ++kk) {                                 // Gauss-Jordan, no pivoting (quasi-definite)
                    if (kk > F && !((kmask >> (kk - F)) & 1ull)) continue;       // pinned unknown: its column is already e_kk
                    const double ipv = frcp(rl_d(Tr[kk], kk));
                    const double fct = (lane == kk) ? 0.0 : Tr[kk] * ipv;
#pragma unroll
                    for (int jj = kk + 1; jj <= m; ++jj) Tr[jj] -= fct * rl_d(Tr[jj], kk);
                }
                double dg = Tr[0];
#pragma unroll
                for (int jj = 1; jj < m; ++jj) if (lane == jj) dg = Tr[jj];
                const double cc_e = (lane < m) ? Tr[m] * frcp(dg) : 0.0;         // lane e: cc[e]
                if (lane < m) L.cc[lane] = cc_e;
                WAVE_LDS_SYNC();
                return cc_e;
            };

            constexpr int NG = (m * m + 63) / 64;
            int gi_[NG], gj_[NG];
#pragma unroll
            for (int s_ = 0; s_ < NG; ++s_) { const int e = lane + 64 * s_; gi_[s_] = e / m; gj_[s_] = e - (e / m) * m; }
            // ---- one structured solve for a whole working set (the ZMP rows in sta[], the kinematic rows in kmask / kact):
            // minimiser u, f and all multipliers; leaves G(W) in L.G and prv / nxt of every row
            auto block_solve = [&](const unsigned long long kmask) __attribute__((always_inline)) {
                // ---- previous / next active row of every row (active or not): exclusive max scan, exclusive suffix min scan
                int nact = 0;
                double cvr[RL];
                // active kinematic rows: right-hand sides sqrt(Qf) (bound_r - (p_r - p_{r-1})) travel to the lanes of their unknowns
                if (klane) L.d1[lane - 1] = (kact != 0) ? sq * ((kact > 0 ? klo : khi) - (L.pf[lane] - L.pf[lane - 1])) : 0.0;
                {
                    int lmax = 0, lmin = 1 << 30;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        const bool act = i <= C && sta[k] != 0;
                        if (act) { lmax = max(lmax, i); lmin = min(lmin, i); }
                        nact += __builtin_popcountll(__builtin_amdgcn_ballot_w64(act));
                        // c_i = bound_i + M_i . plan footsteps
                        cvr[k] = act ? (sta[k] > 0 ? zlo[k] : zhi[k]) + (w1[k] * L.pf[k1[k]] + (1.0 - w1[k]) * L.pf[k1[k] + 1]) : 0.0;
                        if (i <= C) L.sv[i - 1] = cvr[k];                    // c of every row, for its successor
                    }
                    int run = dpp_i<0x138, 0xf>(0, wave_scan_max_i(lmax));
#pragma unroll
                    for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; prv[k] = run; if (i <= C && sta[k] != 0) run = i; }
                    const int rev = __shfl(lmin, 63 - lane);
                    const int ex = dpp_i<0x138, 0xf>(1 << 30, -wave_scan_max_i(-rev));
                    const int nx = __shfl(ex, 63 - lane);
                    run = (nx == (1 << 30)) ? 0 : nx;
#pragma unroll
                    for (int k = RL - 1; k >= 0; --k) { const int i = lane * RL + k + 1; nxt[k] = run; if (i <= C && sta[k] != 0) run = i; }
                }
                WAVE_LDS_SYNC();
                // ---- G = V'K^-1 V / dt^2 and g = V'K^-1 c / dt^2 as sums over consecutive active pairs (p, i) of
                // d d' / gap, d = V_i - V_p.  V_i = Phi(theta_i) + dt PA_i e_E with theta_i the row's mapping weights over the
                // F footstep columns and Phi a fixed sparse map, so everything follows from the Gram sums of
                // [dtheta (F) | dt dPA | dc] weighted by 1 / (dt^2 gap): each lane adds its own rows, one reduction per entry.
                {
                    constexpr int NT = F * (F + 1) / 2, NS = NT + 2 * F + 2;
                    double acc[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) acc[t] = 0.0;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        if (i <= C && sta[k] != 0) {
                            const int p_ = prv[k];
                            int pk1 = -8; double pw1 = 0.0, ppa = 0.0, pc = 0.0;               // V_0 = 0, c_0 = 0
                            if (p_ > 0) { pk1 = L.k1s[p_ - 1]; pw1 = L.w1s[p_ - 1]; ppa = pap[p_]; pc = L.sv[p_ - 1]; }
                            const double pw2 = (p_ > 0) ? 1.0 - pw1 : 0.0, w2 = 1.0 - w1[k];
                            const double om = idt2 * frcp((double)(i - p_));
                            const double dE = dt * (pap[i] - ppa), dc = cvr[k] - pc;
                            double dth[F];
#pragma unroll
                            for (int r = 1; r <= F; ++r) {
                                const double ti = (r == k1[k]) ? w1[k] : ((r == k1[k] + 1) ? w2 : 0.0);
                                const double tp = (r == pk1) ? pw1 : ((r == pk1 + 1) ? pw2 : 0.0);
                                dth[r - 1] = ti - tp;
                            }
                            int t = 0;
#pragma unroll
                            for (int r = 0; r < F; ++r) {
                                const double od = om * dth[r];
#pragma unroll
                                for (int q = r; q < F; ++q) acc[t++] += od * dth[q];
                                acc[NT + r] += od * dE; acc[NT + F + r] += od * dc;
                            }
                            acc[NT + 2 * F] += om * dE * dE; acc[NT + 2 * F + 1] += om * dE * dc;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NS; ++t) { const double v = wave_sum_d(acc[t]); if (lane == 0) L.th[t] = v; }
                    WAVE_LDS_SYNC();
                    // Phi(e): e < F -> +col e+1 ; e > F -> -col (e-F) + col (e-F-1) [if >= 1] ; all scaled by 1/sqrt(Qf)
                    auto TH = [&](int r, int q) -> double {                       // Theta(r, q), 1-based, symmetric
                        const int lo_ = min(r, q), hi_ = max(r, q);
                        return L.th[(lo_ - 1) * F - ((lo_ - 1) * (lo_ - 2)) / 2 + (hi_ - lo_)];
                    };
#pragma unroll
                    for (int s_ = 0; s_ < NG; ++s_) {
                        const int e = lane + 64 * s_;
                        if (e < m * m) {
                            const int i_ = gi_[s_], j_ = gj_[s_];
                            const int ra = i_ < F ? i_ + 1 : i_ - F, rb = j_ < F ? j_ + 1 : j_ - F;     // leading column of Phi(e)
                            const double sa = i_ < F ? 1.0 : -1.0, sb = j_ < F ? 1.0 : -1.0;
                            const bool a2 = i_ > F && ra >= 2, b2 = j_ > F && rb >= 2;                   // second term: +col (r-1)
                            double val;
                            if (i_ == F && j_ == F) val = L.th[NT + 2 * F];
                            else if (i_ == F || j_ == F) {
                                const int r_ = (i_ == F) ? rb : ra; const double s1 = (i_ == F) ? sb : sa; const bool t2 = (i_ == F) ? b2 : a2;
                                val = s1 * L.th[NT + r_ - 1];
                                if (t2) val += L.th[NT + r_ - 2];
                                val *= isq;
                            } else {
                                val = sa * sb * TH(ra, rb);
                                if (a2) val += sb * TH(ra - 1, rb);
                                if (b2) val += sa * TH(ra, rb - 1);
                                if (a2 && b2) val += TH(ra - 1, rb - 1);
                                val *= isq * isq;
                            }
                            L.G[e] = val;
                        }
                    }
                    if (lane < m) {
                        double gv;
                        if (lane == F) gv = L.th[NT + 2 * F + 1] - beq;
                        else {
                            const int ra = lane < F ? lane + 1 : lane - F;
                            gv = (lane < F ? 1.0 : -1.0) * L.th[NT + F + ra - 1];
                            if (lane > F && ra >= 2) gv += L.th[NT + F + ra - 2];
                            gv *= isq;
                        }
                        if (lane > F) gv -= L.d1[lane - F - 1];
                        L.hx[lane] = gv;
                    }
                }
                WAVE_LDS_SYNC();
                (void)solve_small(kmask);
                const double cEw = L.cc[F];
                // comb[r] = (cc[r-1] - ck[r] + ck[r+1]) / sqrt(Qf), r = 1..F: what a row sees through its two footstep columns
                // (ck = the kinematic unknowns, 0 where pinned)
                if (lane <= F + 1) L.comb[lane] = klane ? (L.cc[lane - 1] - L.cc[F + lane] + (lane + 1 <= F ? L.cc[F + lane + 1] : 0.0)) * isq : 0.0;
                if (klane) muK = (kact != 0) ? (kact > 0 ? 1.0 : -1.0) * L.cc[F + lane] : 0.0;
                WAVE_LDS_SYNC();
                double sl[RL];                                               // s_i = c_i - V_i . cc on the active rows
#pragma unroll
                for (int k = 0; k < RL; ++k) {
                    const int i = lane * RL + k + 1;
                    sl[k] = 0.0;
                    if (i <= C && sta[k] != 0)
                        sl[k] = cvr[k] - (w1[k] * L.comb[k1[k]] + (1.0 - w1[k]) * L.comb[k1[k] + 1]) - dt * pap[i] * cEw;
                    if (i <= C) L.sv[i - 1] = sl[k];
                }
                WAVE_LDS_SYNC();
                // multipliers (tridiagonal K^-1), u = dt suffix(lambda) + lambda_E a, f = plan - comb
                double ls = 0.0, suf[RL];
#pragma unroll
                for (int k = RL - 1; k >= 0; --k) {
                    const int i = lane * RL + k + 1;
                    double r_ = 0.0;
                    if (i <= C && sta[k] != 0) {
                        const double sp = prv[k] > 0 ? L.sv[prv[k] - 1] : 0.0;
                        r_ = (sl[k] - sp) * frcp((double)(i - prv[k]));
                        if (nxt[k] > 0) r_ -= (L.sv[nxt[k] - 1] - sl[k]) * frcp((double)(nxt[k] - i));
                        r_ *= idt2;
                    }
                    mu[k] = sta[k] > 0 ? r_ : -r_;
                    ls += dt * r_; suf[k] = ls;
                }
                const double incl = wave_scan_up(ls);
                const double above = rl_d(incl, 63) - incl;
#pragma unroll
                for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; u[k] = (i <= C) ? (suf[k] + above) + cEw * ap[i - 1] : 0.0; }
                if (klane) fr = L.pf[lane] - L.comb[lane];
                muE = cEw;
                qz = nact;
                WAVE_LDS_SYNC();
            };

            // ================= block warm start (primal-dual active-set passes) =================
            // The loop below adds one row per iteration and a nominal tick ends with 40-70 active rows.  Before it, up to
            // c.warm_add passes put every violated ZMP row into the working set at once (and take out rows whose multiplier
            // is not positive), each followed by ONE structured solve for the whole set: G = V'K^-1 V and g = V'K^-1 c from
            // one sweep over the active rows (K^-1 is tridiagonal: gaps only), the (F+1)-unknown system, a tridiagonal apply
            // and a suffix sum.  Up to c.warm_drop more passes only remove rows with negative multipliers.  What is left is a
            // valid starting pair for Goldfarb-Idnani (minimiser on its working set, multipliers >= 0), which finishes the
            // job and owns the kinematic rows; if the passes do not get there the solve starts cold.  Same optimum either way.
            if (c.warm_add > 0) {
                bool cold = false, force_add = false;
                int peel = 1, extra = c.warm_extra, nsolve = 0;
                // closed loop: the working set this instance ended the previous tick with, moved down by one row (the
                // horizon advanced by one sample), is the first guess; any guess is safe, the passes validate it
                int guess[RL];
                bool have_guess = false;
#pragma unroll
                for (int k = 0; k < RL; ++k) guess[k] = 0;
                if (hist != nullptr && hist_load) {
                    const unsigned long long* hq = hist + (size_t)work * 8;
                    unsigned long long any_ = 0ull;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const unsigned long long lo_ = (k < RL - 1) ? hq[k + 1] : (hq[0] >> 1);          // rows on the lower bound
                        const unsigned long long hi_ = (k < RL - 1) ? hq[4 + k + 1] : (hq[4] >> 1);      // rows on the upper bound
                        const int i = lane * RL + k + 1;
                        if (i <= C) guess[k] = ((lo_ >> lane) & 1ull) ? 1 : (((hi_ >> lane) & 1ull) ? -1 : 0);
                        any_ |= lo_ | hi_;
                    }
                    have_guess = any_ != 0ull;
                }
                for (int pass = 0; ; ++pass) {
                    const bool adding = pass < c.warm_add || force_add;
                    force_add = false;
                    PROF_T0();
                    // ---- row values at the current point; the new working set
                    if (lane <= F + 1) L.fl[lane] = fr;
                    WAVE_LDS_SYNC();
                    double lc = 0.0, cm[RL];
#pragma unroll
                    for (int k = 0; k < RL; ++k) { lc += u[k]; cm[k] = lc; }
                    const double bs = wave_scan_up(lc) - lc;
                    // ---- rows that leave: multiplier not positive (while adding) / negative (drop-only passes).  Such a row
                    // usually sits at the end of a run of consecutive rows on the same bound, and the run has to shrink by
                    // more than one row ("peeling"): every pass in a row that still finds one doubles the number of rows
                    // taken off that end (peel).  Taking off too many is harmless, they come back as violated rows.
                    bool xdrop[RL], negr[RL], anyneg = false;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        negr[k] = sta[k] != 0 && (adding ? !(mu[k] > 0.0) : (mu[k] < 0.0));
                        xdrop[k] = false; anyneg = anyneg || negr[k];
                    }
                    const bool wave_neg = __builtin_amdgcn_ballot_w64(anyneg) != 0;
                    if (peel > 1 && wave_neg) {
                        const int sprev = dpp_i<0x138, 0xf>(0, sta[RL - 1]), snext = dpp_i<0x130, 0xf>(0, sta[0]);
                        int lst = 0, len_ = 1 << 30;                            // this lane's last run start / first run end
                        bool isst[RL], isen[RL];
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int i = lane * RL + k + 1;
                            const int sb = k > 0 ? sta[k - 1] : sprev, sa = k < RL - 1 ? sta[k + 1] : snext;
                            isst[k] = sta[k] != 0 && sb != sta[k]; isen[k] = sta[k] != 0 && sa != sta[k];
                            if (isst[k]) lst = i;
                            if (isen[k]) len_ = min(len_, i);
                            if (i <= C) L.sv[i - 1] = negr[k] ? 1.0 : 0.0;
                        }
                        int runlo[RL], runhi[RL];
                        int run = dpp_i<0x138, 0xf>(0, wave_scan_max_i(lst));
#pragma unroll
                        for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; if (isst[k]) run = i; runlo[k] = run; }
                        const int rev = __shfl(len_, 63 - lane);
                        const int ex = dpp_i<0x138, 0xf>(1 << 30, -wave_scan_max_i(-rev));
                        run = __shfl(ex, 63 - lane);
#pragma unroll
                        for (int k = RL - 1; k >= 0; --k) { const int i = lane * RL + k + 1; if (isen[k]) run = i; runhi[k] = run; }
                        WAVE_LDS_SYNC();
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int i = lane * RL + k + 1;
                            if (sta[k] != 0 && runlo[k] >= 1 && runhi[k] <= C && runlo[k] != runhi[k]) {
                                if (i - runlo[k] < peel && L.sv[runlo[k] - 1] != 0.0) xdrop[k] = true;
                                if (runhi[k] - i < peel && L.sv[runhi[k] - 1] != 0.0) xdrop[k] = true;
                            }
                        }
                        WAVE_LDS_SYNC();
                    }
                    peel = wave_neg ? min(2 * peel, 64) : 1;
                    bool changed = false, off_bound = false;
                    double aul = 0.0;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        if (i <= C) {
                            int ns = sta[k];
                            const double v = dt * (cm[k] + bs) - (w1[k] * L.fl[k1[k]] + (1.0 - w1[k]) * L.fl[k1[k] + 1]);
                            aul += ap[i - 1] * u[k];
                            if (have_guess && pass == 0) ns = guess[k];
                            else if (ns != 0) {
                                // an active row must sit on its bound after the block solve; if it does not, the solve broke down
                                const double bd = ns > 0 ? zlo[k] : zhi[k];
                                off_bound = off_bound || !(fabs(v - bd) <= 1e-8 * (fabs(v) + fabs(bd)) + 1e-10);
                                if (negr[k] || xdrop[k]) ns = 0;
                            } else if (adding) {
                                const double tol = 1e-11 * (fabs(v) + fmax(fabs(zlo[k]), fabs(zhi[k]))) + 1e-13;
                                if (v - zlo[k] < -tol) ns = 1; else if (zhi[k] - v < -tol) ns = -1;
                            }
                            changed = changed || ns != sta[k];
                            sta[k] = ns;
                        }
                    }
                    if (nsolve > 0) {
                        const double eqr = wave_sum_d(aul) - beq;                  // ... and the stability row must hold
                        if (__builtin_amdgcn_ballot_w64(off_bound) != 0 || !(fabs(eqr) <= 1e-8 * (1.0 + fabs(beq)))) { cold = true; break; }
                    }
                    if (__builtin_amdgcn_ballot_w64(changed) == 0) {               // a valid pair (and, while adding, nothing violated)
                        if (!adding && extra > 0) { --extra; force_add = true; continue; }   // valid after drop-only passes: one more adding pass
                        if (adding && nsolve > 0) {
                            // every ZMP row was just evaluated at this point (none violated, active ones on their bounds, the
                            // stability row holds, multipliers positive); with the kinematic rows inside their limits this
                            // IS the optimum: skip the Goldfarb-Idnani search and the final re-check
                            const double fprev = dpp64<0x111, 0xf, true>(0.0, fr);
                            bool kbad = false;
                            if (klane && khi < INFINITY) {
                                const double vk = fr - fprev, tol = 1e-11 * (fabs(vk) + fmax(fabs(klo), fabs(khi))) + 1e-13;
                                kbad = !(vk - klo >= -tol && khi - vk >= -tol);
                            }
                            done_opt = __builtin_amdgcn_ballot_w64(kbad) == 0;
                        }
                        break;
                    }
                    if (nsolve >= c.warm_add + c.warm_drop + c.warm_extra * (1 + c.warm_drop)) { cold = true; break; }   // budget spent: start cold
                    ++nsolve; ++iters;
                    block_solve(0ull);                                           // kinematic rows stay out of the block phase
                    PROF(8);
                }
                if (cold) {
                    { PROF_T0(); PROF(9); }
#pragma unroll
                    for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; sta[k] = 0; mu[k] = 0.0; prv[k] = 0; nxt[k] = 0; u[k] = (i <= C) ? t0 * ap[i - 1] : 0.0; }
                    if (klane) fr = L.pf[lane];
                    muE = t0; qz = 0;
                    for (int e = lane; e < m * m; e += 64) L.G[e] = 0.0;
                    WAVE_LDS_SYNC();
                }
            }

            if (!done_opt) for (;;) {
                PROF_T0();
                // ================= most violated inactive row =================
                if (lane <= F + 1) L.fl[lane] = fr;
                WAVE_LDS_SYNC();
                double cand = 0.0, craw = 0.0; int code = 0;
                {
                    double loc = 0.0, cum[RL];
#pragma unroll
                    for (int k = 0; k < RL; ++k) { loc += u[k]; cum[k] = loc; }
                    const double base = wave_scan_up(loc) - loc;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        if (i <= C && sta[k] == 0) {
                            const double v = dt * (cum[k] + base) - (w1[k] * L.fl[k1[k]] + (1.0 - w1[k]) * L.fl[k1[k] + 1]);
                            const double vl = v - zlo[k], vh = zhi[k] - v;
                            const double tol = 1e-11 * (fabs(v) + fmax(fabs(zlo[k]), fabs(zhi[k]))) + 1e-13;
                            if (vl < -tol && vl * inrm[k] < cand) { cand = vl * inrm[k]; craw = vl; code = 2 * i; }
                            if (vh < -tol && vh * inrm[k] < cand) { cand = vh * inrm[k]; craw = vh; code = 2 * i + 1; }
                        }
                    }
                    const double fprev = dpp64<0x111, 0xf, true>(0.0, fr);        // f_{r-1} (lane 0 holds f_0 = 0)
                    if (klane && kact == 0) {
                        const double v = fr - fprev;
                        const double vl = v - klo, vh = khi - v;
                        const double tol = 1e-11 * (fabs(v) + fmax(fabs(klo), fabs(khi))) + 1e-13;
                        if (vl < -tol && vl * knrm < cand) { cand = vl * knrm; craw = vl * sq; code = 2 * (C + lane); }
                        if (vh < -tol && vh * knrm < cand) { cand = vh * knrm; craw = vh * sq; code = 2 * (C + lane) + 1; }
                    }
                }
                const double vmin = wave_min_d(cand);
                if (!(vmin < 0.0)) break;                                         // feasible: done
                const int wl = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(cand == vmin));
                const int cd = rl_i(code, wl);
                double sviol = rl_d(craw, wl);
                const int row = cd >> 1;
                const double sg = (cd & 1) ? -1.0 : 1.0;
                const bool isZ = row <= C;
                const int kr = row - C;                                           // kinematic index when !isZ
                // ---- the new row: border row Vp (one element per lane), footstep part mt, norm, border products dX
                int p_k1 = 0; double p_w1 = 1.0, p_pa = 0.0;
                if (isZ) { p_k1 = L.k1s[row - 1]; p_w1 = L.w1s[row - 1]; p_pa = pap[row]; }
                const double p_w2 = 1.0 - p_w1;
                const double vp = (isZ && lane < m) ? border_elem<F>(lane, p_k1, p_w1, p_pa, dt, isq) : 0.0;
                double mt_e = 0.0, dx_e = 0.0;                                    // lane e: mt[e] (e < F), dX[e] (e >= F)
                if (isZ) { if (lane < F) mt_e = sg * vp; else if (lane < m) dx_e = sg * vp; }
                else {
                    if (lane < F) { const int r = lane + 1; mt_e = (r == kr) ? -sg : ((r == kr - 1) ? sg : 0.0); }      // -sg kvec
                    else if (lane > F && lane < m) { const int r = lane - F; dx_e = sg * ((r == kr) ? (kr >= 2 ? 2.0 : 1.0) : ((r == kr - 1 || r == kr + 1) ? -1.0 : 0.0)); }
                }
                if (lane < m) { L.vp[lane] = vp; L.mt[lane] = mt_e; }
                const double npn = isZ ? (dt * dt * (double)row + ((p_k1 >= 1 ? p_w1 * p_w1 : 0.0) + p_w2 * p_w2) / Qf) : (kr >= 2 ? 2.0 : 1.0);
                double mu_p = 0.0;
                bool failed = false, fresh = true;                                // fresh: sviol still valid from the search
                PROF(0);
                // ================= steps until the row enters (Goldfarb-Idnani) =================
                for (;;) {
                    if (++iters > c.max_iter) { status |= ISMPC_A_ST_ITER_LIMIT; failed = true; break; }
                    // ---- violation of the row at the current point (after a partial step)
                    if (!fresh) {
                        if (lane <= F + 1) L.fl[lane] = fr;
                        WAVE_LDS_SYNC();
                        if (isZ) {
                            double lc = 0.0, cm[RL], vv[RL];
#pragma unroll
                            for (int k = 0; k < RL; ++k) { lc += u[k]; cm[k] = lc; }
                            const double bs = wave_scan_up(lc) - lc;
#pragma unroll
                            for (int k = 0; k < RL; ++k) vv[k] = dt * (cm[k] + bs) - (w1[k] * L.fl[k1[k]] + (1.0 - w1[k]) * L.fl[k1[k] + 1]);
                            const double v = at_row<RL>(vv, row);
                            const double m1 = (p_k1 == 0) ? p_w1 : 0.0;
                            sviol = sg > 0.0 ? v - (zlo0 + m1 * cur) : (zhi0 + m1 * cur) - v;
                        } else {
                            const double fprev = dpp64<0x111, 0xf, true>(0.0, fr);
                            const double vk = sg > 0.0 ? (fr - fprev) - klo : khi - (fr - fprev);
                            sviol = sq * rl_d(vk, kr);
                        }
                    }
                    fresh = false;
                    // ---- neighbours (na < row < nb) of a new ZMP row among the active ones; V there
                    int na = 0, nb = 0; double th = 0.0, va = 0.0, vb = 0.0, vint = 0.0;
                    if (isZ && qz > 0) {
                        na = at_row<RL>(prv, row); nb = at_row<RL>(nxt, row);      // prv / nxt are kept for every row, active or not
                        if (na > 0 && lane < m) va = border_elem<F>(lane, L.k1s[na - 1], L.w1s[na - 1], pap[na], dt, isq);
                        if (nb > 0 && lane < m) vb = border_elem<F>(lane, L.k1s[nb - 1], L.w1s[nb - 1], pap[nb], dt, isq);
                        if (nb == 0) { vint = va; th = 0.0; }
                        else if (na == 0) { th = (double)row * frcp((double)nb); vint = th * vb; }
                        else { th = (double)(row - na) * frcp((double)(nb - na)); vint = va + th * (vb - va); }
                    }
                    // ---- small quasi-definite system  [[I+G11, G1x],[Gx1, Gxx - Sxx]] cc = [h1 ; hx - dX]
                    // unknown order: 0..F-1 footstep columns, F = stability row, F+1..2F = Khat_1..F (inactive: pinned to 0)
                    if (lane < m) {
                        double h_e = sg * vint;
#pragma unroll
                        for (int r = 0; r < F; ++r) h_e += L.G[lane * m + r] * L.mt[r];
                        L.hx[lane] = h_e - dx_e;
                    }
                    WAVE_LDS_SYNC();
                    PROF(1);
                    const unsigned long long kmask = __builtin_amdgcn_ballot_w64(klane && kact != 0);   // bit r: Khat_r active
                    const double cc_e = solve_small(kmask);
                    const double cE = L.cc[F];
                    PROF(2);
                    // ---- y = coefficients on the V columns (delta_Z - V cc = sg dt^2 k_i + V y); rows see the footstep
                    // columns through comb[k1], comb[k1+1]:  comb[r] = (yM_r - yK_r + yK_{r+1}) / sqrt(Qf)
                    if (lane <= F + 1) {
                        double cb = 0.0;
                        if (klane) {
                            const int r = lane;
                            const double yM = L.mt[r - 1] - L.cc[r - 1], yK = -L.cc[F + r], yKn = (r + 1 <= F) ? -L.cc[F + r + 1] : 0.0;
                            cb = (yM - yK + yKn) * isq;
                        }
                        L.comb[lane] = cb;
                    }
                    WAVE_LDS_SYNC();
                    double svl[RL];
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        svl[k] = (i <= C) ? (w1[k] * L.comb[k1[k]] + (1.0 - w1[k]) * L.comb[k1[k] + 1]) - dt * pap[i] * cE : 0.0;
                        if (i <= C) L.sv[i - 1] = svl[k];
                    }
                    WAVE_LDS_SYNC();
                    PROF(4);
                    // ---- rho per active ZMP row (tridiagonal K^-1) + interpolation weights; d.r ; dual step length
                    double rho[RL], ddl = 0.0, tcand = INFINITY; int tcode = 0;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        rho[k] = 0.0;
                        if (i <= C && sta[k] != 0) {
                            const double sp = prv[k] > 0 ? L.sv[prv[k] - 1] : 0.0;
                            double r_ = (svl[k] - sp) * frcp((double)(i - prv[k]));
                            if (nxt[k] > 0) r_ -= (L.sv[nxt[k] - 1] - svl[k]) * frcp((double)(nxt[k] - i));
                            r_ *= idt2;
                            if (isZ) {
                                if (i == na) r_ += (nb == 0) ? sg : sg * (1.0 - th);
                                if (i == nb) r_ += sg * th;
                            }
                            rho[k] = r_;
                            const double w2k = 1.0 - w1[k];
                            double dj;                                             // sg <row+, Z_i>
                            if (isZ) {
                                double mm = 0.0;                                   // M_p . M_i
                                const int a1 = p_k1, b1 = k1[k];
                                if (a1 >= 1) { if (a1 == b1) mm += p_w1 * w1[k]; else if (a1 == b1 + 1) mm += p_w1 * w2k; }
                                { const int cx = a1 + 1; if (cx == b1 && b1 >= 1) mm += p_w2 * w1[k]; else if (cx == b1 + 1) mm += p_w2 * w2k; }
                                dj = sg * (dt * dt * (double)min(row, i) + mm / Qf);
                            } else {
                                double mk = 0.0;                                   // M_i . kvec_kr
                                if (k1[k] == kr) mk += w1[k];
                                if (k1[k] + 1 == kr) mk += w2k;
                                if (kr - 1 >= 1) { if (k1[k] == kr - 1) mk -= w1[k]; if (k1[k] + 1 == kr - 1) mk -= w2k; }
                                dj = sg * (-mk) * isq;
                            }
                            ddl += dj * r_;
                            const double rs = (sta[k] > 0 ? 1.0 : -1.0) * r_;
                            if (rs > 0.0) { const double tt = mu[k] * frcp(rs); if (tt < tcand) { tcand = tt; tcode = i; } }
                        }
                    }
                    if (lane >= F && lane < m) ddl += dx_e * cc_e;                 // border part of d.r
                    const double cK = (klane) ? L.cc[F + lane] : 0.0;              // lane r: unsigned cc of Khat_r
                    if (klane && kact != 0) {
                        const double rs = (kact > 0 ? 1.0 : -1.0) * cK;
                        if (rs > 0.0) { const double tt = muK * frcp(rs); if (tt < tcand) { tcand = tt; tcode = C + lane; } }
                    }
                    const double gamma = npn - wave_sum_d(ddl);
                    const double t1 = wave_min_d(tcand);
                    int lrow = 0;
                    if (t1 < INFINITY) lrow = rl_i(tcode, (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(tcand == t1)));
                    const double t2 = (gamma > 1e-12 * npn) ? -sviol / gamma : INFINITY;
                    const double t = fmin(t1, t2);
                    if (!(t < INFINITY)) { status |= (axis == 0 ? ISMPC_A_ST_X_INFEASIBLE : ISMPC_A_ST_Y_INFEASIBLE); failed = true; break; }
                    PROF(5);
                    // ---- primal step: z_u = suffix sum of (-dt rho, + sg dt at the new row) - r_E a ; z_f from cc
                    if (t2 < INFINITY) {
                        double ls = 0.0, suf[RL];
#pragma unroll
                        for (int k = RL - 1; k >= 0; --k) {
                            const int i = lane * RL + k + 1;
                            ls += -dt * rho[k] + ((isZ && i == row) ? sg * dt : 0.0);
                            suf[k] = ls;                                          // inclusive suffix inside the lane
                        }
                        const double incl = wave_scan_up(ls);
                        const double above = rl_d(incl, 63) - incl;               // lanes above this one
#pragma unroll
                        for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; if (i <= C) u[k] += t * ((suf[k] + above) - cE * ap[i - 1]); }
                        if (klane) {
                            // z_f[r] = ( n+_f[r] + sqrt(Qf) c1[r] - sqrt(Qf) (cK[r] - cK[r+1]) ) / Qf
                            const int r = lane;
                            double nf = isZ ? -sg * L.vp[r - 1] * sq : ((r == kr) ? sg * sq : ((r == kr - 1) ? -sg * sq : 0.0));
                            nf += sq * L.cc[r - 1];
                            nf -= sq * cK;
                            if (r + 1 <= F) nf += sq * L.cc[F + r + 1];
                            fr += t * nf / Qf;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < RL; ++k) if (sta[k] != 0) mu[k] -= t * (sta[k] > 0 ? 1.0 : -1.0) * rho[k];
                    if (klane && kact != 0) muK -= t * (kact > 0 ? 1.0 : -1.0) * cK;
                    muE -= t * cE;
                    mu_p += t;
                    PROF(6);
                    if (t2 < INFINITY && t == t2) {
                        // ============ the row enters ============
                        if (isZ) {
                            if (lane < m) { L.d1[lane] = vp - va; L.d2[lane] = (nb > 0 ? vb : 0.0) - vp; L.d0[lane] = (nb > 0 ? vb : 0.0) - va; }
                            WAVE_LDS_SYNC();
                            const double g1 = idt2 * frcp((double)(row - na)), g2 = nb > 0 ? idt2 * frcp((double)(nb - row)) : 0.0, g0 = nb > 0 ? idt2 * frcp((double)(nb - na)) : 0.0;
                            for (int e = lane; e < m * m; e += 64) {
                                const int i = e / m, jj = e - i * m;
                                L.G[e] += g1 * L.d1[i] * L.d1[jj] + g2 * L.d2[i] * L.d2[jj] - g0 * L.d0[i] * L.d0[jj];
                            }
                            WAVE_LDS_SYNC();
#pragma unroll
                            for (int k = 0; k < RL; ++k) {
                                const int i = lane * RL + k + 1;
                                if (i == row) { sta[k] = sg > 0.0 ? 1 : -1; mu[k] = mu_p; }
                                if (i >= na && i < row) nxt[k] = row;              // rows that now see `row` as their next / previous active row
                                if (i > row && (nb == 0 || i <= nb)) prv[k] = row;
                            }
                            ++qz;
                        } else {
                            if (lane == kr) { kact = sg > 0.0 ? 1 : -1; muK = mu_p; }
                            ++qk;
                        }
                        PROF(7);
                        break;
                    }
                    // ============ partial step: working-set row lrow leaves ============
                    if (lrow <= C) {
                        const int pa_ = at_row<RL>(prv, lrow), pb_ = at_row<RL>(nxt, lrow);
                        double vl_ = 0.0, wa_ = 0.0, wb_ = 0.0;
                        if (lane < m) {
                            vl_ = border_elem<F>(lane, L.k1s[lrow - 1], L.w1s[lrow - 1], pap[lrow], dt, isq);
                            if (pa_ > 0) wa_ = border_elem<F>(lane, L.k1s[pa_ - 1], L.w1s[pa_ - 1], pap[pa_], dt, isq);
                            if (pb_ > 0) wb_ = border_elem<F>(lane, L.k1s[pb_ - 1], L.w1s[pb_ - 1], pap[pb_], dt, isq);
                            L.d1[lane] = vl_ - wa_; L.d2[lane] = (pb_ > 0 ? wb_ : 0.0) - vl_; L.d0[lane] = (pb_ > 0 ? wb_ : 0.0) - wa_;
                        }
                        WAVE_LDS_SYNC();
                        const double g1 = idt2 * frcp((double)(lrow - pa_)), g2 = pb_ > 0 ? idt2 * frcp((double)(pb_ - lrow)) : 0.0, g0 = pb_ > 0 ? idt2 * frcp((double)(pb_ - pa_)) : 0.0;
                        for (int e = lane; e < m * m; e += 64) {
                            const int i = e / m, jj = e - i * m;
                            L.G[e] -= g1 * L.d1[i] * L.d1[jj] + g2 * L.d2[i] * L.d2[jj] - g0 * L.d0[i] * L.d0[jj];
                        }
                        WAVE_LDS_SYNC();
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int i = lane * RL + k + 1;
                            if (i == lrow) { sta[k] = 0; mu[k] = 0.0; }
                            if (i >= pa_ && i < lrow) nxt[k] = pb_;
                            if (i > lrow && (pb_ == 0 || i <= pb_)) prv[k] = pa_;
                        }
                        --qz;
                    } else {
                        if (lane == lrow - C) { kact = 0; muK = 0.0; }
                        --qk;
                    }
                }
                if (failed) break;
            }
            // ---- every row, active or not, the kinematic rows and the stability row are checked once more at the point that
            // is about to be returned: a working set that pins (nearly) every variable can wear the incremental solves down
            // without any inactive row showing it.  One block solve of the final working set (kinematic rows included)
            // polishes such a point; if it still fails, the QP is reported infeasible (the reference's quadprog returns no
            // solution on infeasible QPs).
            if (status == 0 && !done_opt) {
                auto off_point = [&]() __attribute__((always_inline)) -> bool {
                    if (lane <= F + 1) L.fl[lane] = fr;
                    WAVE_LDS_SYNC();
                    double lc = 0.0, cm[RL], aul = 0.0;
#pragma unroll
                    for (int k = 0; k < RL; ++k) { lc += u[k]; cm[k] = lc; }
                    const double bs = wave_scan_up(lc) - lc;
                    bool bad = false;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        if (i <= C) {
                            const double v = dt * (cm[k] + bs) - (w1[k] * L.fl[k1[k]] + (1.0 - w1[k]) * L.fl[k1[k] + 1]);
                            const double tol = 1e-7 * (fabs(v) + fmax(fabs(zlo[k]), fabs(zhi[k]))) + 1e-9;
                            bad = bad || !(v - zlo[k] >= -tol && zhi[k] - v >= -tol);
                            aul += ap[i - 1] * u[k];
                        }
                    }
                    const double fprev = dpp64<0x111, 0xf, true>(0.0, fr);
                    if (klane && khi < INFINITY) {
                        const double v = fr - fprev, tol = 1e-7 * (fabs(v) + fmax(fabs(klo), fabs(khi))) + 1e-9;
                        bad = bad || !(v - klo >= -tol && khi - v >= -tol);
                    }
                    const double eqr = wave_sum_d(aul) - beq;
                    WAVE_LDS_SYNC();
                    return __builtin_amdgcn_ballot_w64(bad) != 0 || !(fabs(eqr) <= 1e-7 * (1.0 + fabs(beq)));
                };
                bool bad = off_point();
                if (bad && c.warm_add > 0) {
                    ++iters;
                    block_solve(__builtin_amdgcn_ballot_w64(klane && kact != 0));
                    double mmax = fabs(muK);
#pragma unroll
                    for (int k = 0; k < RL; ++k) mmax = fmax(mmax, fabs(mu[k]));
                    const double mtol = 1e-8 * (1.0 - wave_min_d(-mmax));
                    bool negm = klane && kact != 0 && muK < -mtol;
#pragma unroll
                    for (int k = 0; k < RL; ++k) negm = negm || (sta[k] != 0 && mu[k] < -mtol);
                    bad = __builtin_amdgcn_ballot_w64(negm) != 0 || off_point();
                }
                if (bad) status |= (axis == 0 ? ISMPC_A_ST_X_INFEASIBLE : ISMPC_A_ST_Y_INFEASIBLE) | ISMPC_A_ST_UNVERIFIED;
            }
        }

#ifdef ISMPC_A_PROF
        { const unsigned long long n_ = __builtin_readcyclecounter(); if (lane == 0 && (work & 127) == 0) { atomicAdd(&g_prof[11], n_ - pq_); atomicAdd(&g_prof[27], 1ull); } pq_ = n_; }
#endif
        if (hist != nullptr) {
            unsigned long long* hq = hist + (size_t)work * 8;
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const unsigned long long lo_ = __builtin_amdgcn_ballot_w64(status == 0 && sta[k] > 0), hi_ = __builtin_amdgcn_ballot_w64(status == 0 && sta[k] < 0);
                if (lane == 0) { hq[k] = lo_; hq[4 + k] = hi_; }
            }
        }
        // ---- LIP update (:297-322), footstep bookkeeping (:522-556), outputs
        const bool ok = (status & (ISMPC_A_ST_BAD_INDEX | ISMPC_A_ST_OVERFLOW)) == 0;
        const double u0 = ok ? rl_d(u[0], 0) : 0.0;
        const double f0 = ok ? rl_d(fr, 1) : cur;
        if (lane == 0) {
            const double p0 = pos, v0 = vel, z0 = zmp;
            double np_, nv_, nz_;
            if (PI) {                                                            // A_upd, B_upd for this instance's eta (:67-71)
                const double ch = cosh(eta * dt), sh = sinh(eta * dt);
                np_ = (ch * p0 + (sh / eta) * v0 + (1 - ch) * z0) + (dt - sh / eta) * u0;
                nv_ = ((eta * sh) * p0 + ch * v0 + (-eta * sh) * z0) + (1 - ch) * u0;
                nz_ = (0.0 * p0 + 0.0 * v0 + 1.0 * z0) + dt * u0;
            } else {
                np_ = (c.Au[0] * p0 + c.Au[1] * v0 + c.Au[2] * z0) + c.Bu[0] * u0;
                nv_ = (c.Au[3] * p0 + c.Au[4] * v0 + c.Au[5] * z0) + c.Bu[1] * u0;
                nz_ = (c.Au[6] * p0 + c.Au[7] * v0 + c.Au[8] * z0) + c.Bu[2] * u0;
            }
            ismpc_a_state* so = state + inst;
            const bool stepped = ok && (j + 1 >= step_ * fc);
            if (ok) {
                if (axis == 0) { so->x = np_; so->xd = nv_; so->xz = nz_; } else { so->y = np_; so->yd = nv_; so->yz = nz_; }
                if (stepped) {
                    const double noff = f0 - fs[fc];
                    if (axis == 0) { so->cur_x = f0; so->off_x = noff; } else { so->cur_y = f0; so->off_y = noff; }
                }
                if (axis == 0) { so->j = j + 1; if (stepped) { so->fc = fc + 1; so->rebuilt = 1; } }
            }
            if (out) {
                ismpc_a_out* o = out + inst;
                const int q = 1 + qz + qk;
                o->com_before[axis] = pos; o->vel_after[axis] = ok ? nv_ : vel; o->u0[axis] = u0; o->f0[axis] = f0;
                if (axis == 0) { o->iters_x = iters; atomicOr(&o->status, status); atomicOr(&o->active, q & 0xffff); }
                else { o->iters_y = iters; atomicOr(&o->status, status); atomicOr(&o->active, (q & 0xffff) << 16); }
            }
        }
        WAVE_LDS_SYNC();
#ifdef ISMPC_A_PROF
        { const unsigned long long n_ = __builtin_readcyclecounter(); if (lane == 0 && (work & 127) == 0) { atomicAdd(&g_prof[12], n_ - pq_); atomicAdd(&g_prof[28], 1ull); } }
#endif
    }
}


// ---- swing-foot re-placement: one thread per instance (closed forms; the 2-/4-variable quadprog is separable, so its
// minimiser is the projection of the target on the box).  trotting/quad_as_bip_no_plots.m:332-426 + compute_two_feet1.m,
// walking/quad_walk_no_plots.m:336-504 + compute_one_feet_walk.m:84-140.
struct FeetParams { int gait, rows; double phi, disp_i, disp_o, disp_forw; };

__device__ __forceinline__ void fixed_diagonal(double fx1, double fy1, double fx2, double fy2, double zx, double zy,
                                               double& m, double& dx, double& dy)
{
    m = (fy2 - fy1) / (fx2 - fx1);
    const double q = fy1 - m * fx1;
    const double xi = (zy + m * zx - q) / (2 * m), yi = m * xi + q;
    dx = zx - xi; dy = zy - yi;
}
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void ismpc_a_feet_kernel(const FeetParams fpz, const ismpc_a_state* __restrict__ prev, const ismpc_a_out* __restrict__ out,
                                    double* __restrict__ feet, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    if (out[b].status & (ISMPC_A_ST_BAD_INDEX | ISMPC_A_ST_OVERFLOW)) return;
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

__global__ void ismpc_a_clear_out(ismpc_a_out* out, int batch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch) { out[i].status = 0; out[i].active = 0; out[i].iters_x = 0; out[i].iters_y = 0; }
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
    bool use_wave = true; int wave_blocks = 0;           // structured wavefront-per-QP kernel (default) vs workgroup-per-QP
    int cus = 0, wave_occ[2] = {0, 0};                   // resident workgroups per CU of the wave kernel (handle-wide / per-instance parameters)
    int* work_counter = nullptr;
    unsigned long long* hist = nullptr; int hist_cap = 0;   // per-QP working set of the previous tick (closed-loop first guess)
    bool hist_ticks = false, hist_valid = false;           // use it in plain tick calls too / it holds the previous tick of this batch
    int hist_batch = 0; bool hist_off = false;            // ISMPC_A_HISTORY=0: never (A/B)
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
    c.warm_add = 4; c.warm_drop = 6; c.warm_extra = 0;    // ISMPC_A_WARM=add,drop,extra overrides; ISMPC_A_WARM=0 starts every QP cold
    if (const char* e = std::getenv("ISMPC_A_WARM")) {
        int a_ = 0, d_ = 0, x_ = 0;
        const int got = std::sscanf(e, "%d,%d,%d", &a_, &d_, &x_);
        if (got >= 1) c.warm_add = std::max(0, std::min(a_, 32));
        if (got >= 2) c.warm_drop = std::max(0, std::min(d_, 32));
        if (got >= 3) c.warm_extra = std::max(0, std::min(x_, 8));
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
    std::vector<double> a(p->C), PA(p->C + 1, 0.0), wt(p->P - p->C);
    double aa = 0.0;
    for (int i = 0; i < p->C; ++i) {
        a[i] = (1 / eta) * (1 - lambda) / (1 - std::pow(lambda, p->C)) * std::exp(-eta * dt * i) - dt * 1.0 * std::exp(-eta * dt * p->C);
        PA[i + 1] = PA[i] + a[i]; aa += a[i] * a[i];
    }
    double sumw = 0.0;
    for (int i = p->C + 1; i <= p->P; ++i) { wt[i - (p->C + 1)] = std::exp(-eta * dt * i) * (1 - std::exp(-eta * dt)); sumw += wt[i - (p->C + 1)]; }
    c.wP = std::exp(-eta * dt * p->P); c.sumw = sumw + c.wP; c.aa = aa;
    h->fsx.resize(p->n_gait); h->fsy.resize(p->n_gait);
    for (int i = 0; i < p->n_gait; ++i) { h->fsx[i] = center[i * 2]; h->fsy[i] = center[i * 2 + 1]; }
    std::vector<double> clx0, cly0, clx1, cly1;
    centreline(h->fsx, p->step, p->ds, p->n_gait, true, clx0);  centreline(h->fsy, p->step, p->ds, p->n_gait, true, cly0);
    centreline(h->fsx, p->step, p->ds, p->n_gait, false, clx1); centreline(h->fsy, p->step, p->ds, p->n_gait, false, cly1);
    c.ncl = (int)std::min(clx0.size(), clx1.size());
    int rc = upload_a(h, a, &c.a);
    if (!rc) rc = upload_a(h, PA, &c.PA);
    if (!rc) rc = upload_a(h, wt, &c.wtail);
    if (!rc) rc = upload_a(h, h->fsx, &c.fsx);
    if (!rc) rc = upload_a(h, h->fsy, &c.fsy);
    if (!rc) { c.plan_x[0] = c.fsx; c.plan_y[0] = c.fsy; c.nplans = 1; c.grav = p->grav; }
    if (!rc) rc = upload_a(h, clx0, &c.clx0);
    if (!rc) rc = upload_a(h, cly0, &c.cly0);
    if (!rc) rc = upload_a(h, clx1, &c.clx1);
    if (!rc) rc = upload_a(h, cly1, &c.cly1);
    if (!rc) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) rc = fail_a(-2, "hipGetDeviceProperties failed");
        else {
            h->slots = prop.multiProcessorCount * (c.sinv_in_lds ? 1 : 4);   // persistent grid: workgroups per CU
            const int scratch_slots = prop.multiProcessorCount * 4;          // the slab always covers the 4-per-CU grid (the LDS variant may fall back to it)
            h->wave_blocks = prop.multiProcessorCount * 4; h->cus = prop.multiProcessorCount;
            if (hipMalloc((void**)&h->work_counter, sizeof(int)) != hipSuccess) rc = fail_a(-3, "counter allocation failed");
            else h->allocs.push_back(h->work_counter);
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
    if (rc) { ismpc_a_destroy(h); return rc; }
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
    if (max_batch > h->hist_cap) {
        if (h->hist) HIP_TRY_A(hipFree(h->hist));
        h->hist = nullptr; h->hist_cap = 0; h->hist_valid = false;
        HIP_TRY_A(hipMalloc((void**)&h->hist, sizeof(unsigned long long) * 16 * (size_t)max_batch));
        h->hist_cap = max_batch;
    }
    return 0;
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
            if (h->hist) HIP_TRY_A(hipFreeAsync(h->hist, s));
            h->hist = nullptr; h->hist_cap = 0; h->hist_valid = false;
            HIP_TRY_A(hipMallocAsync((void**)&h->hist, sizeof(unsigned long long) * 16 * (size_t)batch, s));
            h->hist_cap = batch;
        }
        hist = h->hist;
        if (history == 2 && !(h->hist_valid && h->hist_batch == batch)) history = 1;
        h->hist_valid = true; h->hist_batch = batch;
    }
    const int hist_load = history == 2 ? 1 : 0;
    if (batch > h->prev_cap) {
        if (h->prev) HIP_TRY_A(hipFreeAsync(h->prev, s));
        h->prev = nullptr; h->prev_cap = 0;
        HIP_TRY_A(hipMallocAsync((void**)&h->prev, sizeof(ismpc_a_state) * (size_t)batch, s));
        h->prev_cap = batch;
    }
    HIP_TRY_A(hipMemcpyAsync(h->prev, state_dev, sizeof(ismpc_a_state) * (size_t)batch, hipMemcpyDeviceToDevice, s));
    if (out_dev) hipLaunchKernelGGL(ismpc_a_clear_out, dim3((batch + 255) / 256), dim3(256), 0, s, out_dev, batch);
    if (h->use_wave || inst_dev) {
        // structured solver, one wavefront per QP, 4 per workgroup; persistent grid
        const int rl = (h->c.C + 63) / 64;
        const ismpc_a_state* prev = h->prev;
        const int pi = inst_dev ? 1 : 0;
        HIP_TRY_A(hipMemsetAsync(h->work_counter, 0, sizeof(int), s));
        // persistent grid = exactly the workgroups that are resident at once (registers / LDS decide how many per CU)
#define ISMPC_A_W1(K_) do { if (h->wave_occ[pi] == 0) { int nb = 0; \
                                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, K_, T, 0) != hipSuccess || nb < 1) nb = 1; \
                                h->wave_occ[pi] = nb; } \
                            const int grid = std::min((2 * batch + 3) / 4, h->cus * h->wave_occ[pi]); \
                            hipLaunchKernelGGL(K_, dim3(grid), dim3(T), 0, s, h->c, prev, state_dev, inst_dev, push_dev, out_dev, batch, h->work_counter, hist, hist_load); } while (0)
#define ISMPC_A_W(RL_, F_) do { if (inst_dev) ISMPC_A_W1((ismpc_a_tick_wave<RL_, F_, true>)); else ISMPC_A_W1((ismpc_a_tick_wave<RL_, F_, false>)); } while (0)
#define ISMPC_A_WF(RL_) do { switch (h->c.F) { case 3: ISMPC_A_W(RL_, 3); break; case 4: ISMPC_A_W(RL_, 4); break; \
                                               case 5: ISMPC_A_W(RL_, 5); break; case 6: ISMPC_A_W(RL_, 6); break; default: launched = false; } } while (0)
        bool launched = true;
        switch (rl) { case 1: case 2: ISMPC_A_WF(2); break; case 3: ISMPC_A_WF(3); break; case 4: ISMPC_A_WF(4); break; default: launched = false; }
#undef ISMPC_A_WF
#undef ISMPC_A_W
#undef ISMPC_A_W1
        if (launched) { HIP_TRY_A(hipGetLastError()); return 0; }
        if (inst_dev) return fail_a(-1, "per-instance gait parameters need the structured kernel: 3 <= F <= 6 and C <= 256");
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

#ifdef ISMPC_A_PROF
int ismpc_a_debug_prof(unsigned long long* out32, int reset)
{
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -2;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -2; }
    return 0;
}
#endif

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

int ismpc_a_tick_feet_batch_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, const double* push_dev,
                                   ismpc_a_out* out_dev, double* feet_dev, void* stream)
{
    if (!h || !out_dev || (batch > 0 && !feet_dev) || h->feet.rows == 0) return fail_a(-1, "feet: call ismpc_a_feet_init_device first and pass an output buffer");
    int rc = ismpc_a_tick_batch_device(h, batch, state_dev, push_dev, out_dev, stream);
    if (rc || batch == 0) return rc;
    hipLaunchKernelGGL(ismpc_a_feet_kernel, dim3((batch + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h->feet, (const ismpc_a_state*)h->prev, (const ismpc_a_out*)out_dev, feet_dev, batch);
    HIP_TRY_A(hipGetLastError());
    return 0;
}

int ismpc_a_rollout_feet_device(ismpc_a_handle* h, int batch, ismpc_a_state* state_dev, int ticks, ismpc_a_out* out_traj_dev,
                                double* feet_dev, void* stream)
{
    if (!h || !out_traj_dev || batch < 0 || ticks < 0) return fail_a(-1, "bad argument");
    const bool keep = h->hist_ticks;
    h->hist_ticks = true; h->hist_valid = false;                     // closed loop: previous working set as the first guess
    int rc = 0;
    for (int t = 0; t < ticks && !rc; ++t)
        rc = ismpc_a_tick_feet_batch_device(h, batch, state_dev, nullptr, out_traj_dev + (size_t)t * batch, feet_dev, stream);
    h->hist_ticks = keep; h->hist_valid = false;
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
