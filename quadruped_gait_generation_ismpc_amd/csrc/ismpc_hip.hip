// ISMPC per-tick hot path on gfx950 (MI355X): kernels + the C ABI of include/ismpc.h.
//
// One launch = one MPCSolver::solve (reference AMR_code_DART/MPCSolver.cpp:204-430)
// for every instance of a batch.  A 256-thread workgroup (4 wavefronts) owns
// 16 instances -- the row tile of v_mfma_f64_16x16x4_f64:
//
//   phase A  (wave per instance, lanes = horizon samples)
//            f_z of MPCSolver.cpp:259, with S_bar_z' and S_bar_z_v' applied as
//            suffix sums (they are Toeplitz-triangular, :144-154) -> LDS F[16][NP]
//   phase B  (MFMA)  U = -F * Hinv : the only dense contraction of the tick.
//            Hinv = (q_p S'S + q_v Sv'Sv + q_u I)^-1 is constant (the reference
//            re-forms the Hessian every tick at :258 although it never changes)
//            and shared by the whole batch; B operand streamed from L2.
//   phase C  (wave per instance)
//            - u_i = 0 equalities of :223-243 by a rank-<=F correction
//              (one table column per equality row, chosen by mpcIter)
//            - 0 <= S_bar_z u <= 1e4 check (:158-160), z integration (:274-278)
//            - lambda_j (:296-309), A_j/B_j (:353-361)
//            - phi_state / phi_input (:362-371) as ONE suffix scan of 2x2
//              matrices instead of the reference's O(N^2) cosh/sinh loop
//            - both horizontal QPs (:395-396: H = I, one equality row, a box)
//              solved exactly as continuous quadratic knapsacks
//            - integration (:406-422), 80-byte output record.
//
// There is no CPU fallback in this file: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <new>
#include <algorithm>
#include "ismpc_tables.hpp"
#include "ismpc_sweep.hpp"

// Floating-point contraction is OFF for this file: every fused multiply-add is written as fma().  The same tick arithmetic
// is inlined into several kernels (per-tick, one-launch, in-kernel rollout, resume) whose results must agree bit for bit,
// and implicit contraction is a per-context optimiser decision.
#pragma clang fp contract(off)

namespace {

constexpr int TI = 16;          // instances per workgroup = MFMA M tile

typedef double d4 __attribute__((ext_vector_type(4)));

struct DevConst {
    int N, NP, NPs, S, F, nmid, npat, Fmax, rows, tick_divisor;
    double dt, cdt, mass, g, h_des, half_run, half_first, q_p, q_u, q_v, z_lo, z_hi, gate, eta;
    double inv_mass, dt_over_mass, inv_eta, sim_div, cdt_over_dt;   // 1/m, dt/m (B_z), 1/eta (C_sc), dt/cdt, cdt/dt: uniform divisions hoisted to the host
    const double *Hinv, *W, *midx, *midy, *midz, *tailx, *taily, *ftsp_t;
    const int *e_lo, *ne;
    // affine form of the vertical stage (ismpc_tables.hpp)
    const double *vtab, *tz, *tg, *dU, *SdU, *Wt, *SW;
    int flat;
    // inequality fallback (0 <= S u <= 1e4 active)
    const double *HSt, *SHSt;
    const DevConst* sets; int nsets;  // parameter sweeps (ismpc_create_sweep): one record per parameter set, its own tables and scalars; the
                                      // instance's record names its set (ismpc_tick_in.reserved).  NULL / 0 for a plain handle
    const int* order;                 // sweeps, after ismpc_sweep_bind: the instances of the bound batch sorted by parameter set.  Slot g of the
                                      // launch runs instance order[g], so the lane groups of a wavefront read ONE set's tables, and workgroup b
                                      // takes the slots of virtual block sweep_vblock(b): the workgroups an XCD receives (b mod 8) cover one
                                      // contiguous eighth of the sorted batch -- K / 8 sets' tables per L2 instead of all K.  NULL: slot g = instance g
    int* zflag;                       // four self-resetting counters (zeroed once, at ismpc_create): [0] entries in the deferred list of the
                                      // running two-launch step, [1] fallback workgroups done with it, [2] instances an in-kernel rollout
                                      // parked for its resume launch, [3] resume workgroups done.  The consumer launch exits at once on a
                                      // zero count; otherwise its LAST workgroup zeroes the pair again -- so the counters are valid whatever
                                      // launched before (a rollout between two ticks, hipGraph replays of one captured step: the count does
                                      // not depend on launch ids and no memset sits outside a captured step)
    int* zseen;                       // id of the last launch that deferred an instance, in a word of host memory (written, never read, by the device): how the host picks the launch form
    double* zpool; int* zbusy;        // active-set fallback: slots of zstride doubles (G^-1 cap x cap + per-entry vectors), one lock word per slot
    int zslots, zcap, zldsq; size_t zstride;   // zldsq: entries the fallback keeps in its LDS window before it moves to a slot (Z_LDS_Q; ISMPC_Z_LDS_Q lowers it: tests)
    // sample-major copies for ismpc_tick_quad: a lane's R samples are one contiguous run (16-byte loads, one base address)
    const double *vq;                 // (npat+1) x NT x 6 : U0,Ua,Ub,SU0,SUa,SUb per sample
    const double *tzg;                // NT x 2 : tz, tg per sample
    const double *midxy;              // nmid x 2 : midx, midy per sample
    // the same tables laid out for the lane-group kernels' shape (R samples per lane, LPI lanes per instance), so that one
    // wave-wide load instruction reads LPI x 16 contiguous bytes per instance (a lane's samples are NOT contiguous here):
    const double *vqT;                // (npat+1) x R x 3 x LPI double2 : pair k of sample li*R + r at [((p R + r) 3 + k) LPI + li]
    const double *tzgT;               // R x LPI double2 : (tz, tg) of sample li*R + r at [r LPI + li]
};

// ---- wavefront (64 lanes) primitives: DPP, no LDS crossbar (ds_bpermute) on the critical path ----
// DPP controls (GFX9 / CDNA): row_shl:n = 0x100+n, row_shr:n = 0x110+n, wave_shl:1 = 0x130,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143.  A "row" is 16 lanes.
template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ double dpp64(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
    return __hiloint2double(hi, lo);
}
template <int LANE>
__device__ __forceinline__ double readlane64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), LANE), __builtin_amdgcn_readlane(__double2loint(v), LANE));
}
// inclusive prefix sum over lanes 0..lane
__device__ __forceinline__ double wave_scan_up(double v)
{
    v += dpp64<0x111, 0xf, true>(0.0, v);
    v += dpp64<0x112, 0xf, true>(0.0, v);
    v += dpp64<0x114, 0xf, true>(0.0, v);
    v += dpp64<0x118, 0xf, true>(0.0, v);
    v += dpp64<0x142, 0xa, false>(0.0, v);      // rows 1,3 += lane 15 of the row below
    v += dpp64<0x143, 0xc, false>(0.0, v);      // rows 2,3 += lane 31
    return v;
}
// wave-uniform sum of all 64 lanes
__device__ __forceinline__ double wave_sum(double v) { return readlane64<63>(wave_scan_up(v)); }
// sum over lanes strictly below this one
__device__ __forceinline__ double wave_prefix_excl(double v) { return wave_scan_up(v) - v; }
__device__ __forceinline__ double bcast0(double v) { return readlane64<0>(v); }

struct M2 { double a, b, c, d; };   // [a b; c d]
__device__ __forceinline__ M2 mul(const M2& x, const M2& y)
{
    M2 r;
    r.a = fma(x.a, y.a, x.b * y.c); r.b = fma(x.a, y.b, x.b * y.d);
    r.c = fma(x.c, y.a, x.d * y.c); r.d = fma(x.c, y.b, x.d * y.d);
    return r;
}
template <int CTRL>
__device__ __forceinline__ M2 dpp_m2_ident(const M2& y)     // out-of-range source lane -> identity
{
    M2 t;
    t.a = dpp64<CTRL, 0xf, false>(1.0, y.a); t.b = dpp64<CTRL, 0xf, false>(0.0, y.b);
    t.c = dpp64<CTRL, 0xf, false>(0.0, y.c); t.d = dpp64<CTRL, 0xf, false>(1.0, y.d);
    return t;
}
template <int LANE>
__device__ __forceinline__ M2 readlane_m2(const M2& y)
{
    return (M2){readlane64<LANE>(y.a), readlane64<LANE>(y.b), readlane64<LANE>(y.c), readlane64<LANE>(y.d)};
}
// X_lane = Y_63 Y_62 ... Y_{lane+1} (identity for lane 63); total = Y_63 ... Y_0
__device__ __forceinline__ M2 wave_suffix_product_excl(M2 y, int lane, M2& total)
{
    y = mul(dpp_m2_ident<0x101>(y), y);       // row_shl:1  (lane L reads lane L+1 of its row)
    y = mul(dpp_m2_ident<0x102>(y), y);
    y = mul(dpp_m2_ident<0x104>(y), y);
    y = mul(dpp_m2_ident<0x108>(y), y);
    // first lane of each 16-lane row now holds that row's product; fold the rows above in
    const M2 p1 = readlane_m2<16>(y), p2 = readlane_m2<32>(y), p3 = readlane_m2<48>(y);
    const M2 m1 = mul(p3, p2), m0 = mul(m1, p1);
    const int row = lane >> 4;
    M2 pre = (M2){1.0, 0.0, 0.0, 1.0};
    if (row == 2) pre = p3; else if (row == 1) pre = m1; else if (row == 0) pre = m0;
    y = mul(pre, y);
    total = readlane_m2<0>(y);
    return dpp_m2_ident<0x130>(y);            // wave_shl:1 -> exclusive
}

// sinh(x)/x and (cosh(x)-1)/x^2 as functions of w = x^2.  Taylor to w^7 is exact to < 1 ulp for
// w <= 0.25 (next term 4e-20); beyond that (lambda dt^2 > 0.25: never on a physical gait) libm.
__device__ __forceinline__ void sinhc_coshc(double w, double& P, double& Q)
{
    if (__builtin_expect(w <= 0.25, 1)) {
        P = 1.0 / 1307674368000.0;                 // 1/15!
        P = fma(P, w, 1.0 / 6227020800.0);            // 1/13!
        P = fma(P, w, 1.0 / 39916800.0);              // 1/11!
        P = fma(P, w, 1.0 / 362880.0);                // 1/9!
        P = fma(P, w, 1.0 / 5040.0);                  // 1/7!
        P = fma(P, w, 1.0 / 120.0);                   // 1/5!
        P = fma(P, w, 1.0 / 6.0);                     // 1/3!
        P = fma(P, w, 1.0);
        Q = 1.0 / 20922789888000.0;                // 1/16!
        Q = fma(Q, w, 1.0 / 87178291200.0);           // 1/14!
        Q = fma(Q, w, 1.0 / 479001600.0);             // 1/12!
        Q = fma(Q, w, 1.0 / 3628800.0);               // 1/10!
        Q = fma(Q, w, 1.0 / 40320.0);                 // 1/8!
        Q = fma(Q, w, 1.0 / 720.0);                   // 1/6!
        Q = fma(Q, w, 1.0 / 24.0);                    // 1/4!
        Q = fma(Q, w, 0.5);
    } else {
        const double x = sqrt(w);
        P = sinh(x) / x;
        Q = (cosh(x) - 1.0) / w;
    }
}

// Caller bookkeeping in front of solve(): Controller.cpp:297-304 (enabled) and :310.
// Per-launch scratch of the inequality fallback of the two-launch form: the LIST of deferred instances (batch ints).  The per-tick
// kernel appends an instance under the handle's counter DevConst::zflag[0]; the fallback launch behind it walks exactly those entries
// (scanning 65 536 marks with 256 wavefronts cost 0.5 ms whenever anything was deferred) and its last workgroup zeroes the counter.
__host__ __device__ inline size_t zscratch_bytes(int batch) { return 4 * (size_t)batch + 16; }
__device__ __forceinline__ int* zlist_of(unsigned char* zmark, int) { return reinterpret_cast<int*>(zmark); }
struct Walk { double sim; int mpc, ctl, fc; };
__device__ __forceinline__ Walk load_walk(const DevConst& c, const ismpc_tick_in* rec, int rollout_frame)
{
    Walk w; w.sim = rec->simulation_time; w.mpc = rec->mpc_iter; w.ctl = rec->control_iter; w.fc = rec->footstep_counter;
    if (rollout_frame >= 0) {
        if (w.fc >= 0 && w.fc < c.rows && w.sim >= c.ftsp_t[w.fc] - 1) { w.ctl = 0; w.mpc = 0; w.fc = w.fc + 1; }
        w.sim = (double)rollout_frame;
    }
    return w;
}
// 0 = run the tick, else the pass-through status (MPCSolver.cpp:214; index range of :259,381)
__device__ __forceinline__ int gate_tick(const DevConst& c, const Walk& w, int& idx)
{
    idx = 0;
    if ((w.ctl % c.tick_divisor) != 0) return ISMPC_ST_TICK_SKIPPED;
    const double t = (c.sim_div == 1.0) ? w.sim : w.sim / c.sim_div;
    if (!(t > -1.0) || !(t < 2.0e9)) return ISMPC_ST_BAD_INDEX;
    idx = (int)t;
    if (idx < 0 || idx + 2 * c.N > c.nmid || w.mpc < 0) return ISMPC_ST_BAD_INDEX;
    return 0;
}

// R = horizon samples per lane (N <= 64 R); WAVES = wavefronts per workgroup (16 instances per
// workgroup either way, each wavefront walks TI / WAVES of them through phases A and C).
template <int R, int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void ismpc_tick_dense(const DevConst c, const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                       ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj, int batch, int rollout_frame)
{
    constexpr int IPW = TI / WAVES;
    extern __shared__ double smem[];                  // [TI][NPs]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int inst0 = blockIdx.x * TI;
    const int N = c.N, NP = c.NP, NPs = c.NPs;
    const double dt = c.dt;
    const ismpc_tick_in* in = (rollout_frame >= 0) ? state_rw : in_ro;

    // ---------------- phase A: f_z, MPCSolver.cpp:259 ----------------
    // lanes hold the horizon REVERSED here (lane L <-> samples (63-L) R ..): S_bar_z' and S_bar_z_v'
    // are sums over LATER samples, which this way are prefix sums over lanes (DPP row_shr / row_bcast).
    for (int q = 0; q < IPW; ++q) {
        const int li = wave * IPW + q;
        const int gi = inst0 + li;
        const int nb = (63 - lane) * R;
        double f[R];
#pragma unroll
        for (int r = 0; r < R; ++r) f[r] = 0.0;
        if (gi < batch) {
            const Walk w = load_walk(c, in + gi, rollout_frame);
            int idx;
            if (gate_tick(c, w, idx) == 0) {
                const double z = in[gi].com_pos[2], zd = in[gi].com_vel[2];
                double rp[R], rv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int n = nb + r;
                    if (n < N) {
                        const double k = (double)n;
                        // T_bar_z(k,:) s + T_bar_g_z(k) - h_des - mid_z ; T_bar_z_v(k,:) s + T_bar_g_z_v(k)
                        rp[r] = (z + (k + 1.0) * dt * zd) - c.g * dt * dt * (0.5 * k * (k + 1.0)) - c.h_des - c.midz[idx + n];
                        rv[r] = zd - c.g * dt * k;
                    } else { rp[r] = 0.0; rv[r] = 0.0; }
                }
                // T_j = sum_{k>=j} rp_k ;  V_i = sum_{j>i} T_j = sum_{k>i} (k-i) rp_k ;  TV_i = sum_{k>i} rv_k
                double tp[R], lp = 0.0, lv = 0.0, tv[R];
#pragma unroll
                for (int r = R - 1; r >= 0; --r) { tv[r] = lv; lv += rv[r]; lp += rp[r]; tp[r] = lp; }
                const double up = wave_prefix_excl(lp);
                const double uv = wave_prefix_excl(lv);
                double vt[R], lt = 0.0;
#pragma unroll
                for (int r = R - 1; r >= 0; --r) { tp[r] += up; vt[r] = lt; lt += tp[r]; }
                const double ut = wave_prefix_excl(lt);
                const double cs = dt * dt / c.mass, cv = dt / c.mass;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int n = nb + r;
                    if (n < N) f[r] = c.q_p * cs * (vt[r] + ut) + c.q_v * cv * (tv[r] + uv) - c.q_u * c.mass * c.g;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { const int n = nb + r; if (n < NP) smem[li * NPs + n] = f[r]; }
    }
    __syncthreads();

    // ---------------- phase B: U = -F Hinv on the matrix cores ----------------
    {
        constexpr int MAXT = (16 + WAVES - 1) / WAVES;   // NP <= 256 -> at most 16 column tiles
        const int ntiles = NP >> 4;
        d4 acc[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        const int arow = lane & 15, kq = lane >> 4;
        if (wave < ntiles) {
            for (int kk = 0; kk < NP; kk += 4) {
                const double a = smem[arow * NPs + kk + kq];                   // A[i = lane&15][k = lane>>4]
                const double* brow = c.Hinv + (size_t)(kk + kq) * NP + arow;   // B[k = lane>>4][j = lane&15]
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    const int tile = wave + t * WAVES;
                    if (tile < ntiles) {
                        const double b = brow[tile * 16];
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                               // every wave is done reading F
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int tile = wave + t * WAVES;
            if (tile < ntiles) {
#pragma unroll
                for (int v = 0; v < 4; ++v)            // D: col = lane&15, row = (lane>>4) + 4*v
                    smem[(kq + 4 * v) * NPs + tile * 16 + arow] = -acc[t][v];
            }
        }
    }
    __syncthreads();

    // ---------------- phase C: everything after the vertical solve ----------------
    for (int q = 0; q < IPW; ++q) {
        const int li = wave * IPW + q;
        const int gi = inst0 + li;
        if (gi >= batch) continue;
        const ismpc_tick_in* rec = in + gi;
        const Walk w = load_walk(c, rec, rollout_frame);
        const double x0 = rec->com_pos[0], y0 = rec->com_pos[1], z0 = rec->com_pos[2];
        const double xd0 = rec->com_vel[0], yd0 = rec->com_vel[1], zd0 = rec->com_vel[2];
        int idx;
        int status = gate_tick(c, w, idx);
        double o_x = x0, o_y = y0, o_z = z0, o_xd = xd0, o_yd = yd0, o_zd = zd0;
        double uz0 = 0.0, ux0 = 0.0, uy0 = 0.0;
        int itx = 0, ity = 0;
        double u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = 0.0;
        double tau[2] = {0.0, 0.0}, sgx = 1.0, sgy = 1.0, hbox = 0.0;
        double a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = 0.0;
        bool stage3 = false;

        if (status == 0) {
            // ---- stage 1 tail: equality correction (MPCSolver.cpp:223-243, is_running :262-263)
#pragma unroll
            for (int r = 0; r < R; ++r) { const int n = lane * R + r; u[r] = (n < N) ? smem[li * NPs + n] : 0.0; }
            if (w.fc > 1 && w.mpc < c.npat) {
                const int elo = c.e_lo[w.mpc], ne = c.ne[w.mpc];
                const double* Wp = c.W + (size_t)w.mpc * c.Fmax * NP;
                for (int e = 0; e < ne; ++e) {
                    const double ue = smem[li * NPs + elo + e];
#pragma unroll
                    for (int r = 0; r < R; ++r) { const int n = lane * R + r; if (n < N) u[r] -= Wp[(size_t)e * NP + n] * ue; }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) { const int n = lane * R + r; if (n >= elo && n < elo + ne) u[r] = 0.0; }
            }
            // ---- S_bar_z u = (dt^2/m) * exclusive prefix of inclusive prefix of u
            double ci[R], lc = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) { lc += u[r]; ci[r] = lc; }
            const double pc = wave_prefix_excl(lc);
            double di[R], ld_ = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) { ci[r] += pc; di[r] = ld_; ld_ += ci[r]; }
            const double pd = wave_prefix_excl(ld_);
            const double cs = dt * dt / c.mass;
            bool viol = false;
            double lam[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = lane * R + r;
                const double k = (double)n;
                const double su = cs * (di[r] + pd);
                if (n < N && (su < c.z_lo - 1e-11 * fmax(1.0, fabs(c.z_lo)) || su > c.z_hi + 1e-11 * fmax(1.0, fabs(c.z_hi)))) viol = true;   // beyond rounding
                const double zpos = su + (z0 + (k + 1.0) * dt * zd0) - c.g * dt * dt * (0.5 * k * (k + 1.0));
                const double zacc = (1.0 / c.mass) * u[r] - c.g;
                lam[r] = (c.g + zacc) / zpos;                               // MPCSolver.cpp:306
            }
            if (__builtin_amdgcn_ballot_w64(viol) != 0) status |= ISMPC_ST_Z_INEQ_ACTIVE;
            uz0 = bcast0(u[0]);
            // ---- z integration, MPCSolver.cpp:274-278
            o_z = z0 + dt * zd0;
            o_zd = zd0 + (dt / c.mass) * uz0 - dt * c.g;
            if (isnan(o_z)) { o_z = c.h_des; status |= ISMPC_ST_Z_NAN; }
            if (isnan(o_zd)) { o_zd = 0.0; status |= ISMPC_ST_Z_NAN; }

            // ---- A_j, B_j per sample, MPCSolver.cpp:353-361, in the form
            //   A = [1 + wQ, dt P; lambda dt P, 1 + wQ],  B = [-wQ, -lambda dt P],  w = lambda dt^2,
            //   P = sinh(x)/x, Q = (cosh(x)-1)/x^2, x = sqrt(lambda) dt: no sqrt, no division, and
            //   lambda < gate (A = [1 dt; 0 1], B = 0) is simply lambda := 0.
            M2 A[R]; double B0[R], B1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = lane * R + r;
                const double le = (lam[r] < c.gate) ? 0.0 : lam[r];
                const double dtn = (n < N) ? dt : 0.0;
                const double wv = le * dtn * dtn;
                double P, Q;
                sinhc_coshc(wv, P, Q);
                const double ch1 = wv * Q, s1 = dtn * P, s2 = le * s1;
                A[r] = (M2){1.0 + ch1, s1, s2, 1.0 + ch1};
                B0[r] = -ch1; B1[r] = -s2;
            }
            const double lam0 = bcast0(lam[0]);
            const M2 A0 = readlane_m2<0>(A[0]);
            const double B00 = bcast0(B0[0]), B10 = bcast0(B1[0]);

            if (lam0 > c.gate) {                                           // MPCSolver.cpp:322
                stage3 = true;
                // ---- suffix products: X_lane = A_{N-1} ... A_{first sample of lane+1}
                M2 Y = A[0];
#pragma unroll
                for (int r = 1; r < R; ++r) Y = mul(A[r], Y);
                M2 tot;
                const M2 X = wave_suffix_product_excl(Y, lane, tot);
                // row vector c_n = C_sc A_{N-1} ... A_{n+1},  C_sc = [1, 1/eta]  (MPCSolver.cpp:375-379)
                const double ie = 1.0 / c.eta;
                double c0 = X.a + ie * X.c, c1 = X.b + ie * X.d;
#pragma unroll
                for (int r = R - 1; r >= 0; --r) {
                    a[r] = c0 * B0[r] + c1 * B1[r];                        // Aeq(n) = C_sc phi_input(:,n)
                    const double n0 = c0 * A[r].a + c1 * A[r].c, n1 = c0 * A[r].b + c1 * A[r].d;
                    c0 = n0; c1 = n1;
                }
                const double cps0 = tot.a + ie * tot.c, cps1 = tot.b + ie * tot.d;   // C_sc phi_state
                // ---- box midpoints and reductions
                const double h = (w.fc > 1) ? c.half_run : c.half_first;     // MPCSolver.cpp:328-338
                hbox = h;
                double aa[R];
                double s_abs = 0.0, s_ax = 0.0, s_ay = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int n = lane * R + r;
                    double mx = 0.0, my = 0.0;
                    if (n < N) { mx = c.midx[idx + n]; my = c.midy[idx + n]; } else a[r] = 0.0;
                    aa[r] = fabs(a[r]);
                    s_abs += aa[r]; s_ax += a[r] * mx; s_ay += a[r] * my;
                }
                s_abs = wave_sum(s_abs); s_ax = wave_sum(s_ax); s_ay = wave_sum(s_ay);
                const double beq_x = -(cps0 * x0 + cps1 * xd0) + c.tailx[idx];   // MPCSolver.cpp:381-384
                const double beq_y = -(cps0 * y0 + cps1 * yd0) + c.taily[idx];
                // v = u - mid:  sum a v = bp,  |v| <= h   ->  v_n = sg * sign(a_n) * min(tau |a_n|, h);
                // G(tau) = sum |a_n| min(tau |a_n|, h) is concave piecewise linear: Newton from tau = 0 is
                // monotone and lands on the exact breakpoint interval in a handful of steps.
                const double bpx = beq_x - s_ax, bpy = beq_y - s_ay;
                sgx = (bpx < 0.0) ? -1.0 : 1.0; sgy = (bpy < 0.0) ? -1.0 : 1.0;
                const double T[2] = { fabs(bpx), fabs(bpy) };
                const double gmax = h * s_abs;
                bool done[2]; int prev[2] = {-1, -1}, its[2] = {0, 0};
#pragma unroll
                for (int ax = 0; ax < 2; ++ax) {
                    const bool inf = T[ax] > gmax * (1.0 + 1e-12) + 1e-300;
                    if (inf) { status |= (ax == 0 ? ISMPC_ST_X_INFEASIBLE : ISMPC_ST_Y_INFEASIBLE); tau[ax] = INFINITY; }
                    done[ax] = inf;
                }
                for (int it = 0; it < N + 2 && !(done[0] && done[1]); ++it) {
                    double ssat[2] = {0.0, 0.0}, qfree[2] = {0.0, 0.0}; int cnt[2] = {0, 0};
#pragma unroll
                    for (int ax = 0; ax < 2; ++ax) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const bool sat = tau[ax] * aa[r] >= h;
                            ssat[ax] += sat ? aa[r] : 0.0;
                            qfree[ax] += sat ? 0.0 : a[r] * a[r];
                            cnt[ax] += __popcll(__builtin_amdgcn_ballot_w64(sat));
                        }
                    }
#pragma unroll
                    for (int ax = 0; ax < 2; ++ax) {
                        if (done[ax]) continue;
                        if (cnt[ax] == prev[ax]) { done[ax] = true; continue; }
                        const double ss = wave_sum(ssat[ax]), qf = wave_sum(qfree[ax]);
                        ++its[ax];
                        if (!(qf > 0.0)) { tau[ax] = INFINITY; done[ax] = true; continue; }
                        const double tn = (T[ax] - h * ss) / qf;
                        if (!(tn > tau[ax])) { done[ax] = true; continue; }
                        tau[ax] = tn; prev[ax] = cnt[ax];
                    }
                }
                itx = its[0]; ity = its[1];
                {   // first decision variables (lane 0 holds sample 0)
                    const double a0 = bcast0(a[0]), aa0 = fabs(a0), sa0 = (a0 < 0.0) ? -1.0 : 1.0;
                    const double m0x = c.midx[idx], m0y = c.midy[idx];
                    ux0 = m0x + sgx * sa0 * ((aa0 > 0.0) ? fmin(tau[0] * aa0, h) : 0.0);
                    uy0 = m0y + sgy * sa0 * ((aa0 > 0.0) ? fmin(tau[1] * aa0, h) : 0.0);
                }
            } else {
                status |= ISMPC_ST_FLIGHT;
            }
            // ---- integration with A(lambda_0), B(lambda_0), MPCSolver.cpp:406-422
            o_x  = (A0.a * x0 + A0.b * xd0) + B00 * ux0;
            o_xd = (A0.c * x0 + A0.d * xd0) + B10 * ux0;
            o_y  = (A0.a * y0 + A0.b * yd0) + B00 * uy0;
            o_yd = (A0.c * y0 + A0.d * yd0) + B10 * uy0;
        }

        // ---- 80-byte output record: lanes 0..9 store one 8-byte word each
        {
            double word = 0.0;
            const long long packed = (long long)(unsigned)status | ((long long)(unsigned)((itx & 255) | ((ity & 255) << 8)) << 32);
            switch (lane) {
                case 0: word = o_x; break;  case 1: word = o_y; break;  case 2: word = o_z; break;
                case 3: word = o_xd; break; case 4: word = o_yd; break; case 5: word = o_zd; break;
                case 6: word = uz0; break;  case 7: word = ux0; break;  case 8: word = uy0; break;
                case 9: word = __longlong_as_double(packed); break;
                default: break;
            }
            if (out && lane < 10) reinterpret_cast<double*>(out + gi)[lane] = word;
        }
        if (u_traj) {
            double* dst = u_traj + (size_t)gi * 3 * N;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = lane * R + r;
                if (n < N) {
                    double vx = 0.0, vy = 0.0;
                    if (stage3) {
                        const double aa = fabs(a[r]), sa = (a[r] < 0.0) ? -1.0 : 1.0;
                        vx = c.midx[idx + n] + sgx * sa * ((aa > 0.0) ? fmin(tau[0] * aa, hbox) : 0.0);
                        vy = c.midy[idx + n] + sgy * sa * ((aa > 0.0) ? fmin(tau[1] * aa, hbox) : 0.0);
                    }
                    dst[n] = u[r]; dst[N + n] = vx; dst[2 * N + n] = vy;
                }
            }
        }
        // ---- closed loop: feed back (Controller.cpp:346-348) and advance counters (:503-504)
        if (rollout_frame >= 0 && lane == 0) {
            ismpc_tick_in* st = state_rw + gi;
            st->com_pos[0] = o_x; st->com_pos[1] = o_y; st->com_pos[2] = o_z;
            st->com_vel[0] = o_xd; st->com_vel[1] = o_yd; st->com_vel[2] = o_zd;
            st->simulation_time = w.sim;
            const int ctl = w.ctl + 1;
            st->control_iter = ctl;
            st->mpc_iter = (int)floor(ctl * c.cdt / c.dt);
            st->footstep_counter = w.fc;
        }
    }
}

// ======================================================================================
// Fast path: one wavefront = one instance, no LDS, no barrier.
// The vertical QP (MPCSolver.cpp:220-278) is evaluated from the affine tables (the dense solve
// happened once at ismpc_create); what is left per tick is the nonlinear part: lambda_j, the
// 2x2 suffix scan, and the two exact knapsack solves.
// ======================================================================================
template <int CTRL>
__device__ __forceinline__ double dpp64z(double src)        // DPP move, out-of-range source lanes read 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// y <- T y for T = I + Tm taken from another lane (Tm = 0 where that lane does not exist)
template <int CTRL>
__device__ __forceinline__ void scan_step(M2& y)
{
    const double ta = dpp64z<CTRL>(y.a - 1.0), tb = dpp64z<CTRL>(y.b), tc = dpp64z<CTRL>(y.c), td = dpp64z<CTRL>(y.d - 1.0);
    M2 r;
    r.a = fma(ta, y.a, fma(tb, y.c, y.a)); r.b = fma(ta, y.b, fma(tb, y.d, y.b));
    r.c = fma(tc, y.a, fma(td, y.c, y.c)); r.d = fma(tc, y.b, fma(td, y.d, y.d));
    y = r;
}
template <int R> __device__ __forceinline__ void loadR(const double* p, double (&v)[R])
{
    if constexpr (R == 2) { const double2 t = *reinterpret_cast<const double2*>(p); v[0] = t.x; v[1] = t.y; }
    else if constexpr (R == 4) { const double2 t = *reinterpret_cast<const double2*>(p), q = *reinterpret_cast<const double2*>(p + 2); v[0] = t.x; v[1] = t.y; v[2] = q.x; v[3] = q.y; }
    else {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = p[r];
    }
}

// ---- vertical QP with active inequality rows (MPCSolver.cpp:158-160: 0 <= S_bar_z u <= 1e4), rare path ----
// Dual active-set (Goldfarb-Idnani step logic) in range-space form over the inequality rows only: the equalities are
// already inside the reduced inverse P_p = (I - W_p E_p') Hinv, so with p_k = P_p S_k' and g_k = S p_k (rows of the HSt /
// SHSt tables, pattern folded in with Wt / SW) the Gram matrix of the working set is G[j][k] = g_k[row_j].
// The working set may grow to every row of the horizon (the reference's solver, utils.cpp:264-383, has no cap either), so
// G^-1 (q x q) and the per-entry vectors live in a slot of a handle-owned pool in HBM; one wavefront owns a slot while it
// solves.  Nothing here is on the hot path: the nominal and perturbed gait workloads never activate a row.
__device__ __forceinline__ double readlane_dyn(double v, int l)
{
    const int ll = __builtin_amdgcn_readfirstlane(l);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), ll), __builtin_amdgcn_readlane(__double2loint(v), ll));
}
__device__ __forceinline__ int readlane_dyn(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ double wave_allmax(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_allmin(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_allmin_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
template <int R>
__device__ __forceinline__ double sample_at(const double (&v)[R], int k)     // v at sample k (k wave-uniform)
{
    const int owner = k / R, slot = k - owner * R;
    double x = v[0];
#pragma unroll
    for (int r = 1; r < R; ++r) if (slot == r) x = v[r];
    return readlane_dyn(x, owner);
}
// Folds the equality pattern into a row of (HSt, SHSt) or into a combination of such rows: p -= W_e ue_e, g -= (S W)_e ue_e with
// ue_e = the UNPROJECTED vector at the e-th pinned sample (lane e holds it in `uel`); the pinned samples of p end up zero.
template <int R>
__device__ __forceinline__ void z_project(const DevConst& c, int n0, int pat, int elo, int ne, double uel, double (&pc)[R], double (&gc)[R])
{
    constexpr int NT = ismpc::Tables::NT;
    for (int e = 0; e < ne; ++e) {                       // (not unrolled: the lane read is a convergent operation)
        const double ue = readlane_dyn(uel, e);
        double wv[R], sv[R];
        loadR<R>(c.Wt + ((size_t)pat * c.Fmax + e) * NT + n0, wv); loadR<R>(c.SW + ((size_t)pat * c.Fmax + e) * NT + n0, sv);
#pragma unroll
        for (int r = 0; r < R; ++r) { pc[r] = fma(-wv[r], ue, pc[r]); gc[r] = fma(-sv[r], ue, gc[r]); }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { const int n = n0 + r; if (n >= elo && n < elo + ne) pc[r] = 0.0; }
}
template <int R>
__device__ __forceinline__ void z_fetch(const DevConst& c, int lane, int row, int n0, int pat, int elo, int ne, double (&pc)[R], double (&gc)[R])
{
    constexpr int NT = ismpc::Tables::NT;
    loadR<R>(c.HSt + (size_t)row * NT + n0, pc); loadR<R>(c.SHSt + (size_t)row * NT + n0, gc);
    const double uel = (lane < ne) ? c.HSt[(size_t)row * NT + elo + lane] : 0.0;
    z_project<R>(c, n0, pat, elo, ne, uel, pc, gc);
}
// one wavefront's stores to its working storage become visible to its other lanes (same CU: a wait for the stores is all it takes)
#define Z_MEMSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

// A slot of the pool: lane 0 takes the first free one starting at `hint`.  Holders always finish (bounded iteration
// count) and wait for nobody, so spinning here cannot deadlock, whatever is resident.
__device__ __forceinline__ int z_slot_acquire(const DevConst& c, int lane, int hint)
{
    int s = 0;
    if (lane == 0) {
        s = (int)((unsigned)hint % (unsigned)c.zslots);
        while (atomicCAS(&c.zbusy[s], 0, 1) != 0) { s = (s + 1 == c.zslots) ? 0 : s + 1; __builtin_amdgcn_s_sleep(8); }
    }
    s = __builtin_amdgcn_readfirstlane(s);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return s;
}
__device__ __forceinline__ void z_slot_release(const DevConst& c, int lane, int slot)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) atomicExch(&c.zbusy[slot], 0);
}

// Working storage of one solve: G^-1 (ld x ld), the per-entry vectors, g of the entering row by sample, the entries' rows.
// Up to Z_LDS_Q entries it is the wavefront's own LDS window (Z_LDS_DOUBLES doubles, handed in by the kernel); a working set
// that outgrows it moves to a slot of the pool in HBM (ld = zcap) and stays there.  The pointers are wave-uniform and generic.
constexpr int Z_LDS_Q = 16;
constexpr int Z_LDS_DOUBLES = Z_LDS_Q * Z_LDS_Q + 4 * Z_LDS_Q + ismpc::Tables::NT + Z_LDS_Q / 2;
struct ZStore {
    double *Ginv, *amu, *asg, *rv, *dv, *gs; int* arow; int ld;
    __device__ __forceinline__ void bind(double* base, int ld_)
    {
        Ginv = base; ld = ld_; amu = base + (size_t)ld_ * ld_; asg = amu + ld_; rv = asg + ld_; dv = rv + ld_; gs = dv + ld_;
        arow = reinterpret_cast<int*>(gs + ismpc::Tables::NT);
    }
};

// returns the iteration count; updates u, su in place.  Entry j of the working set: row arow[j], bound sign asg[j] (+1 lower,
// -1 upper), multiplier amu[j]; Ginv = G^-1 over the entries.
template <int R>
__device__ int z_active_set(const DevConst& c, int lane, int n0, int pat, double (&u)[R], double (&su)[R], int& status, int slot_hint, double* lds)
{
    constexpr int NT = ismpc::Tables::NT;
    const int N = c.N, cap = c.zcap;
    int elo = 0, ne = 0;
    if (pat < c.npat) { elo = c.e_lo[pat]; ne = c.ne[pat]; }
    const double tol_lo = 1e-11 * fmax(1.0, fabs(c.z_lo)), tol_hi = 1e-11 * fmax(1.0, fabs(c.z_hi));
    ZStore z; z.bind(lds, c.zldsq);
    int slot = -1;
    bool sact[R];
#pragma unroll
    for (int r = 0; r < R; ++r) sact[r] = false;
    int q = 0, its = 0;
    const int max_its = 8 * N + 64;
    Z_MEMSYNC();                                             // whatever the caller kept in the window has been read
    for (;;) {
        // ---- most violated free row
        double best = 0.0; int code = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r;
            if (n < N && !sact[r]) {
                const double vl = c.z_lo - su[r], vh = su[r] - c.z_hi;
                if (vl > tol_lo && vl > best) { best = vl; code = 2 * n; }
                if (vh > tol_hi && vh > best) { best = vh; code = 2 * n + 1; }
            }
        }
        const double vmax = wave_allmax(best);
        if (!(vmax > 0.0)) break;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(best == vmax);
        code = readlane_dyn(code, (int)__builtin_ctzll(m));
        const int row = code >> 1;
        const double sg = (code & 1) ? -1.0 : 1.0;
        if (q >= cap) { status |= ISMPC_ST_Z_FAILED; break; }             // cannot happen: entries are distinct rows, cap = N
        if (q == z.ld && slot < 0) {
            // ---- the working set outgrows the LDS window: everything moves to a pool slot
            slot = z_slot_acquire(c, lane, slot_hint);
            ZStore zp; zp.bind(c.zpool + (size_t)slot * c.zstride, cap);
#pragma nounroll
            for (int j = lane; j < q; j += 64) {
#pragma nounroll
                for (int k = 0; k < q; ++k) zp.Ginv[(size_t)k * cap + j] = z.Ginv[k * z.ld + j];
                zp.amu[j] = z.amu[j]; zp.asg[j] = z.asg[j]; zp.arow[j] = z.arow[j];
            }
            z = zp;
            Z_MEMSYNC();
        }
        double* const Ginv = z.Ginv; double* const amu = z.amu; double* const asg = z.asg; double* const rv = z.rv; double* const dv = z.dv;
        double* const gs = z.gs; int* const arow = z.arow;
        const size_t ld = (size_t)z.ld;
        double pc[R], gc[R];
        z_fetch<R>(c, lane, row, n0, pat, elo, ne, pc, gc);
        const double npn = sample_at<R>(gc, row);
#pragma unroll
        for (int r = 0; r < R; ++r) if (n0 + r < NT) gs[n0 + r] = gc[r];
        Z_MEMSYNC();
        double mu_p = 0.0;
        bool fail = false;
        for (;;) {
            if (++its > max_its) { fail = true; break; }
            const double srow = sample_at<R>(su, row);
            const double sviol = sg > 0.0 ? srow - c.z_lo : c.z_hi - srow;
            // d_j = sg * asg_j * g_row[arow_j] ;  r = G^-1 d ;  ratio test over the entries
#pragma nounroll
            for (int j = lane; j < q; j += 64) dv[j] = sg * asg[j] * gs[arow[j]];
            Z_MEMSYNC();
            double drl = 0.0, tcl = INFINITY; int tl = 1 << 30;
#pragma nounroll
            for (int j = lane; j < q; j += 64) {
                double acc = 0.0;
#pragma unroll 4
                for (int k = 0; k < q; ++k) acc = fma(Ginv[(size_t)k * ld + j], dv[k], acc);        // column j = row j (symmetric)
                rv[j] = acc; drl = fma(dv[j], acc, drl);
                if (acc > 0.0) { const double tt = amu[j] / acc; if (tt < tcl) { tcl = tt; tl = j; } }
            }
            Z_MEMSYNC();
            const double gamma = npn - wave_sum(drl);
            const double t1 = wave_allmin(tcl);
            const double t2 = (gamma > 1e-12 * npn) ? -sviol / gamma : INFINITY;
            const double t = fmin(t1, t2);
            if (!(t < INFINITY)) { fail = true; break; }
            if (t2 < INFINITY) {
                // z = P (n+ - N r): the entries' rows of (HSt, SHSt) combined with coefficient -r_j asg_j (independent loads, four
                // in flight), the equality pattern folded into the combination ONCE (it is linear), plus the new row
                double zu[R], zs[R], vu[R], vs[R];
#pragma unroll
                for (int r = 0; r < R; ++r) { vu[r] = 0.0; vs[r] = 0.0; }
                double uel = 0.0;
#pragma unroll 4
                for (int j = 0; j < q; ++j) {
                    const double cf = -rv[j] * asg[j];
                    const size_t rj = (size_t)arow[j] * NT;
                    double pj[R], gj[R];
                    loadR<R>(c.HSt + rj + n0, pj); loadR<R>(c.SHSt + rj + n0, gj);
                    const double uj = (lane < ne) ? c.HSt[rj + elo + lane] : 0.0;
                    uel = fma(cf, uj, uel);
#pragma unroll
                    for (int r = 0; r < R; ++r) { vu[r] = fma(cf, pj[r], vu[r]); vs[r] = fma(cf, gj[r], vs[r]); }
                }
                if (q > 0) z_project<R>(c, n0, pat, elo, ne, uel, vu, vs);
#pragma unroll
                for (int r = 0; r < R; ++r) { zu[r] = fma(sg, pc[r], vu[r]); zs[r] = fma(sg, gc[r], vs[r]); }
#pragma unroll
                for (int r = 0; r < R; ++r) { u[r] = fma(t, zu[r], u[r]); su[r] = fma(t, zs[r], su[r]); }
            }
#pragma nounroll
            for (int j = lane; j < q; j += 64) amu[j] -= t * rv[j];
            mu_p += t;
            if (t2 < INFINITY && t == t2) {
                // ---- the row enters: border update of G^-1
                const double ig = 1.0 / gamma;
#pragma nounroll
                for (int j = lane; j < q; j += 64) {
                    const double rj = rv[j];
#pragma unroll 4
                    for (int k = 0; k < q; ++k) Ginv[(size_t)k * ld + j] = fma(rv[k] * ig, rj, Ginv[(size_t)k * ld + j]);
                    Ginv[(size_t)q * ld + j] = -rj * ig; Ginv[(size_t)j * ld + q] = -rj * ig;
                }
                if (lane == 0) { Ginv[(size_t)q * ld + q] = ig; arow[q] = row; asg[q] = sg; amu[q] = mu_p; }
#pragma unroll
                for (int r = 0; r < R; ++r) if (n0 + r == row) sact[r] = true;
                ++q;
                Z_MEMSYNC();
                break;
            }
            // ---- entry l (the first one attaining t1) leaves: Schur update, the last entry moves into its place
            Z_MEMSYNC();
            const int l = wave_allmin_i((tcl == t1) ? tl : (1 << 30));
            const int last = q - 1;
            const int drow = arow[l];
            const double piv = Ginv[(size_t)l * ld + l];
#pragma nounroll
            for (int j = lane; j < q; j += 64) rv[j] = Ginv[(size_t)l * ld + j];                    // column l
            Z_MEMSYNC();
#pragma nounroll
            for (int j = lane; j < q; j += 64) {
                if (j == l) continue;
                const double cj = rv[j] / piv;
#pragma nounroll
                for (int k = 0; k < q; ++k) if (k != l) Ginv[(size_t)k * ld + j] -= rv[k] * cj;
            }
            Z_MEMSYNC();
            if (l != last) {
#pragma nounroll
                for (int j = lane; j < q; j += 64) {
                    if (j == l) continue;
                    const double vl_ = Ginv[(size_t)last * ld + j];
                    Ginv[(size_t)l * ld + j] = vl_; Ginv[(size_t)j * ld + l] = vl_;
                }
                Z_MEMSYNC();
                if (lane == 0) { Ginv[(size_t)l * ld + l] = Ginv[(size_t)last * ld + last]; arow[l] = arow[last]; asg[l] = asg[last]; amu[l] = amu[last]; }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) if (n0 + r == drow) sact[r] = false;
            --q;
            Z_MEMSYNC();
        }
        if (fail) { status |= ISMPC_ST_Z_FAILED; break; }
    }
    if (slot >= 0) z_slot_release(c, lane, slot);
    Z_MEMSYNC();                                             // the window is the caller's again
    return its;
}

// 1/x to rounding error: v_rcp_f64 + two Newton steps (the IEEE division sequence is about three times as long)
__device__ __forceinline__ double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

template <int R, bool FB>
__device__ __forceinline__ void tick_affine_body(const DevConst& c, const int gi, const int lane,
                                                 const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                                                 ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj,
                                                 int rollout_frame, unsigned char* zmark, int launch_id, int* zlist = nullptr, int zbatch = 0, double* zlds = nullptr,
                                                 const int extra_status = 0)
{
    constexpr int NT = ismpc::Tables::NT;
    const int N = c.N;
    bool deferred = false;                            // FB == false: an instance with active inequality rows is left to the fallback kernel
    const double dt = c.dt;
    const ismpc_tick_in* rec = ((rollout_frame >= 0) ? state_rw : in_ro) + gi;
    const Walk w = load_walk(c, rec, rollout_frame);
    const double x0 = rec->com_pos[0], y0 = rec->com_pos[1], z0 = rec->com_pos[2];
    const double xd0 = rec->com_vel[0], yd0 = rec->com_vel[1], zd0 = rec->com_vel[2];
    int idx;
    int status = gate_tick(c, w, idx) | extra_status;       // (extra_status: a sweep instance that names no parameter set -- passed through)
    double o_x = x0, o_y = y0, o_z = z0, o_xd = xd0, o_yd = yd0, o_zd = zd0;
    double uz0 = 0.0, ux0 = 0.0, uy0 = 0.0;
    int itx = 0, ity = 0, zits = 0;
    const int n0 = lane * R;                          // this lane owns samples n0 .. n0+R-1 (tables are zero past N)
    double u[R], a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { u[r] = 0.0; a[r] = 0.0; }
    double tau0 = 0.0, tau1 = 0.0, sgx = 1.0, sgy = 1.0, hbox = 0.0;
    bool stage3 = false;

    if (status == 0) {
        // ---- vertical stage from the affine tables; pattern = which u_i = 0 rows are present
        // (MPCSolver.cpp:223-243, is_running :262-263)
        const int pat = (w.fc > 1 && w.mpc < c.npat) ? w.mpc : c.npat;
        const double* T = c.vtab + (size_t)pat * 6 * NT + n0;
        double t0[R], t1[R], t2[R], su[R], tz[R], tg[R];
        loadR<R>(T, t0); loadR<R>(T + NT, t1); loadR<R>(T + 2 * NT, t2);
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = fma(zd0, t2[r], fma(z0, t1[r], t0[r]));
        loadR<R>(T + 3 * NT, t0); loadR<R>(T + 4 * NT, t1); loadR<R>(T + 5 * NT, t2);
#pragma unroll
        for (int r = 0; r < R; ++r) su[r] = fma(zd0, t2[r], fma(z0, t1[r], t0[r]));
        loadR<R>(c.tz + n0, tz); loadR<R>(c.tg + n0, tg);
        if (!c.flat) {                                  // plans with mid_z != 0 (MPCSolver.cpp:259)
            double du[R], ds[R];
            loadR<R>(c.dU + (size_t)idx * NT + n0, du); loadR<R>(c.SdU + (size_t)idx * NT + n0, ds);
            int elo = 0, ne = 0;
            if (pat < c.npat) { elo = c.e_lo[pat]; ne = c.ne[pat]; }
            for (int e = 0; e < ne; ++e) {
                const double ue = c.dU[(size_t)idx * NT + elo + e];
                double wv[R], sv[R];
                loadR<R>(c.Wt + ((size_t)pat * c.Fmax + e) * NT + n0, wv); loadR<R>(c.SW + ((size_t)pat * c.Fmax + e) * NT + n0, sv);
#pragma unroll
                for (int r = 0; r < R; ++r) { du[r] = fma(-wv[r], ue, du[r]); ds[r] = fma(-sv[r], ue, ds[r]); }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = n0 + r;
                u[r] += du[r]; su[r] += ds[r];
                if (n >= elo && n < elo + ne) u[r] = 0.0;
            }
        }
        bool viol = false;
        double lam[R];
        const double zlo_t = c.z_lo - 1e-11 * fmax(1.0, fabs(c.z_lo)), zhi_t = c.z_hi + 1e-11 * fmax(1.0, fabs(c.z_hi));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r;
            viol = viol || (n < N && (su[r] < zlo_t || su[r] > zhi_t));         // MPCSolver.cpp:158-160, beyond rounding
        }
        const bool anyviol = __builtin_amdgcn_ballot_w64(viol) != 0;
        if (anyviol) {
            status |= ISMPC_ST_Z_INEQ_ACTIVE;
            if constexpr (FB) zits = z_active_set<R>(c, lane, n0, pat, u, su, status, gi, zlds);
            else deferred = true;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double zpos = su[r] + fma(tz[r], zd0, z0) + tg[r];            // S u + T_bar_z s + T_bar_g_z
            const double zacc = fma(c.inv_mass, u[r], -c.g);
            lam[r] = (c.g + zacc) * frcp(zpos);                                 // MPCSolver.cpp:306
        }
        uz0 = bcast0(u[0]);
        o_z = fma(dt, zd0, z0);                                                 // MPCSolver.cpp:274-278
        o_zd = fma(c.dt_over_mass, uz0, zd0) - dt * c.g;
        if (isnan(o_z)) { o_z = c.h_des; status |= ISMPC_ST_Z_NAN; }
        if (isnan(o_zd)) { o_zd = 0.0; status |= ISMPC_ST_Z_NAN; }

        // ---- A_j, B_j (MPCSolver.cpp:353-361): A = [1+wQ, dt P; lam dt P, 1+wQ], B = [-wQ, -lam dt P]
        double ch1[R], s1[R], s2[R];
        bool big = false, mid = false;
        double wv_[R], le_[R], dtn_[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r;
            le_[r] = (lam[r] < c.gate) ? 0.0 : lam[r];
            dtn_[r] = (n < N) ? dt : 0.0;
            wv_[r] = le_[r] * dtn_[r] * dtn_[r];
            big = big || (wv_[r] > 0.25);
            mid = mid || (wv_[r] > 0.004);
        }
        // w = lambda dt^2 is <= 0.0025 on a physical gait (lambda <= 25 at dt = 0.01): degree 3 in w is then exact to
        // < 1 ulp (next term w^4/9! <= 7e-16 relative to 1 at w = 0.004); the wave takes degree 7 only if some lane needs it
        if (__builtin_amdgcn_ballot_w64(mid) == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double wv = wv_[r];
                double P = 1.0 / 5040.0, Q = 1.0 / 40320.0;
                P = fma(P, wv, 1.0 / 120.0);         Q = fma(Q, wv, 1.0 / 720.0);
                P = fma(P, wv, 1.0 / 6.0);           Q = fma(Q, wv, 1.0 / 24.0);
                P = fma(P, wv, 1.0);                 Q = fma(Q, wv, 0.5);
                ch1[r] = wv * Q; s1[r] = dtn_[r] * P; s2[r] = le_[r] * s1[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double wv = wv_[r];
                double P = 1.0 / 1307674368000.0, Q = 1.0 / 20922789888000.0;
                P = fma(P, wv, 1.0 / 6227020800.0);  Q = fma(Q, wv, 1.0 / 87178291200.0);
                P = fma(P, wv, 1.0 / 39916800.0);    Q = fma(Q, wv, 1.0 / 479001600.0);
                P = fma(P, wv, 1.0 / 362880.0);      Q = fma(Q, wv, 1.0 / 3628800.0);
                P = fma(P, wv, 1.0 / 5040.0);        Q = fma(Q, wv, 1.0 / 40320.0);
                P = fma(P, wv, 1.0 / 120.0);         Q = fma(Q, wv, 1.0 / 720.0);
                P = fma(P, wv, 1.0 / 6.0);           Q = fma(Q, wv, 1.0 / 24.0);
                P = fma(P, wv, 1.0);                 Q = fma(Q, wv, 0.5);
                ch1[r] = wv * Q; s1[r] = dtn_[r] * P; s2[r] = le_[r] * s1[r];
            }
        }
        if (__builtin_amdgcn_ballot_w64(big) != 0) {      // lambda dt^2 > 1/4: off any physical gait; libm, wave-uniform
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = n0 + r;
                const double le = (lam[r] < c.gate) ? 0.0 : lam[r];
                const double dtn = (n < N) ? dt : 0.0;
                const double wv = le * dtn * dtn;
                if (wv > 0.25) { const double x = sqrt(wv); ch1[r] = cosh(x) - 1.0; s1[r] = dtn * (sinh(x) / x); s2[r] = le * s1[r]; }
            }
        }
        const double lam0 = bcast0(lam[0]);
        const double A0a = 1.0 + bcast0(ch1[0]), A0b = bcast0(s1[0]), A0c = bcast0(s2[0]);

        if (lam0 > c.gate) {                                                    // MPCSolver.cpp:322
            stage3 = true;
            // ---- inclusive suffix product over lanes: Y_L = A(block 63) ... A(block L), row by row
            M2 Y = (M2){1.0 + ch1[0], s1[0], s2[0], 1.0 + ch1[0]};
#pragma unroll
            for (int r = 1; r < R; ++r) Y = mul((M2){1.0 + ch1[r], s1[r], s2[r], 1.0 + ch1[r]}, Y);
            scan_step<0x101>(Y); scan_step<0x102>(Y); scan_step<0x104>(Y); scan_step<0x108>(Y);   // row_shl 1,2,4,8
            // g_row = C_sc P_3 .. P_{row+1}  (P_r = product of row r = Y at its first lane), C_sc = [1, 1/eta]
            const double ie = c.inv_eta;
            const M2 p1 = readlane_m2<16>(Y), p2 = readlane_m2<32>(Y), p3 = readlane_m2<48>(Y);
            const double g2a = fma(ie, p3.c, p3.a), g2b = fma(ie, p3.d, p3.b);
            const double g1a = fma(g2b, p2.c, g2a * p2.a), g1b = fma(g2b, p2.d, g2a * p2.b);
            const double g0a = fma(g1b, p1.c, g1a * p1.a), g0b = fma(g1b, p1.d, g1a * p1.b);
            const int row = lane >> 4;
            const double ga = row == 3 ? 1.0 : (row == 2 ? g2a : (row == 1 ? g1a : g0a));
            const double gb = row == 3 ? ie  : (row == 2 ? g2b : (row == 1 ? g1b : g0b));
            // cv_L = C_sc (suffix product from the first sample of lane L) ; the lane needs it one lane up
            const double cva = fma(gb, Y.c, ga * Y.a), cvb = fma(gb, Y.d, ga * Y.b);
            const double cps0 = readlane64<0>(cva), cps1 = readlane64<0>(cvb);  // C_sc phi_state
            double c0 = dpp64<0x130, 0xf, false>(1.0, cva), c1 = dpp64<0x130, 0xf, false>(ie, cvb);   // wave_shl:1
            // ---- Aeq(n) = C_sc phi_input(:,n) = c_n B_n, walking the lane's samples backwards
#pragma unroll
            for (int r = R - 1; r >= 0; --r) {
                a[r] = -fma(c0, ch1[r], c1 * s2[r]);
                const double k0 = fma(c0, ch1[r], fma(c1, s2[r], c0)), k1 = fma(c1, ch1[r], fma(c0, s1[r], c1));
                c0 = k0; c1 = k1;
            }
            const double h = (w.fc > 1) ? c.half_run : c.half_first;            // MPCSolver.cpp:328-338
            hbox = h;
            double q0 = 0.0, s_ax = 0.0, s_ay = 0.0;
            double mx[R], my[R];                          // loaded here, not earlier: 8 waves per SIMD hide the latency, registers are the scarce resource
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = n0 + r;
                mx[r] = (n < N) ? c.midx[idx + n] : 0.0; my[r] = (n < N) ? c.midy[idx + n] : 0.0;
            }
            const double tailx = c.tailx[idx], taily = c.taily[idx];
#pragma unroll
            for (int r = 0; r < R; ++r) { q0 = fma(a[r], a[r], q0); s_ax = fma(a[r], mx[r], s_ax); s_ay = fma(a[r], my[r], s_ay); }
            q0 = wave_sum(q0); s_ax = wave_sum(s_ax); s_ay = wave_sum(s_ay);
            const double bpx = (tailx - fma(cps0, x0, cps1 * xd0)) - s_ax;      // beq - a'mid, MPCSolver.cpp:381-384
            const double bpy = (taily - fma(cps0, y0, cps1 * yd0)) - s_ay;
            sgx = (bpx < 0.0) ? -1.0 : 1.0; sgy = (bpy < 0.0) ? -1.0 : 1.0;
            // min 1/2|v|^2, a'v = bp, |v| <= h  ->  v_n = sg sign(a_n) min(tau |a_n|, h): Newton on the concave
            // piecewise-linear G(tau) = sum |a_n| min(tau |a_n|, h) from tau = 0 (first step: tau = |bp| / sum a^2)
            const double T[2] = { fabs(bpx), fabs(bpy) };
            const double iq0 = frcp(q0);
            double tau[2] = { T[0] * iq0, T[1] * iq0 };
            int its[2] = {1, 1};
            double aa[R];
#pragma unroll
            for (int r = 0; r < R; ++r) aa[r] = fabs(a[r]);
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
                if (!(q0 > 0.0)) {                                               // no sample can move the ZMP
                    tau[ax] = (T[ax] > 0.0) ? INFINITY : 0.0;
                    if (T[ax] > 1e-300) status |= (ax == 0 ? ISMPC_ST_X_INFEASIBLE : ISMPC_ST_Y_INFEASIBLE);
                }
                int prev = 0;
                for (int it = 0; it < N + 2; ++it) {
                    int cnt = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) cnt += __popcll(__builtin_amdgcn_ballot_w64(tau[ax] * aa[r] >= h));
                    if (cnt == prev) break;                                      // active set unchanged: exact
                    double ssat = 0.0, qfree = 0.0;
#pragma unroll
                    for (int r = 0; r < R; ++r) { const bool sat = tau[ax] * aa[r] >= h; ssat += sat ? aa[r] : 0.0; const double a2 = a[r] * a[r]; qfree += sat ? 0.0 : a2; }
                    ssat = wave_sum(ssat); qfree = wave_sum(qfree);
                    ++its[ax];
                    const double rem = fma(-h, ssat, T[ax]);
                    if (!(qfree > 0.0)) {                                        // everything saturated
                        if (rem > fma(h * ssat, 1e-12, 1e-300)) status |= (ax == 0 ? ISMPC_ST_X_INFEASIBLE : ISMPC_ST_Y_INFEASIBLE);
                        tau[ax] = INFINITY; break;
                    }
                    const double tn = rem * frcp(qfree);
                    if (!(tn > tau[ax])) break;
                    tau[ax] = tn; prev = cnt;
                }
            }
            tau0 = tau[0]; tau1 = tau[1]; itx = its[0]; ity = its[1];
            {   // first decision variables (lane 0 holds sample 0)
                const double a0 = bcast0(a[0]), aa0 = fabs(a0), sa0 = (a0 < 0.0) ? -1.0 : 1.0;
                const double m0x = bcast0(mx[0]), m0y = bcast0(my[0]);
                ux0 = fma(sgx * sa0, (aa0 > 0.0) ? fmin(tau0 * aa0, h) : 0.0, m0x);
                uy0 = fma(sgy * sa0, (aa0 > 0.0) ? fmin(tau1 * aa0, h) : 0.0, m0y);
            }
        } else {
            status |= ISMPC_ST_FLIGHT;
        }
        // ---- integration with A(lambda_0), B(lambda_0), MPCSolver.cpp:406-422
        o_x  = fma(1.0 - A0a, ux0, fma(A0a, x0, A0b * xd0));
        o_xd = fma(-A0c, ux0, fma(A0c, x0, A0a * xd0));
        o_y  = fma(1.0 - A0a, uy0, fma(A0a, y0, A0b * yd0));
        o_yd = fma(-A0c, uy0, fma(A0c, y0, A0a * yd0));
    }

    // ---- 80-byte output record: lanes 0..9 store one 8-byte word each
    {
        double word = 0.0;
        const long long packed = (long long)(unsigned)status | ((long long)(unsigned)((itx & 255) | ((ity & 255) << 8) | ((zits & 255) << 16)) << 32);
        switch (lane) {
            case 0: word = o_x; break;  case 1: word = o_y; break;  case 2: word = o_z; break;
            case 3: word = o_xd; break; case 4: word = o_yd; break; case 5: word = o_zd; break;
            case 6: word = uz0; break;  case 7: word = ux0; break;  case 8: word = uy0; break;
            case 9: word = __longlong_as_double(packed); break;
            default: break;
        }
        if (out && lane < 10) reinterpret_cast<double*>(out + gi)[lane] = word;
    }
    if (u_traj) {
        double* dst = u_traj + (size_t)gi * 3 * N;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r;
            if (n < N) {
                double vx = 0.0, vy = 0.0;
                if (stage3) {
                    const double aa = fabs(a[r]), sa = (a[r] < 0.0) ? -1.0 : 1.0;
                    vx = fma(sgx * sa, (aa > 0.0) ? fmin(tau0 * aa, hbox) : 0.0, c.midx[idx + n]);
                    vy = fma(sgy * sa, (aa > 0.0) ? fmin(tau1 * aa, hbox) : 0.0, c.midy[idx + n]);
                }
                dst[n] = u[r]; dst[N + n] = vx; dst[2 * N + n] = vy;
            }
        }
    }
    // ---- closed loop: feed back (Controller.cpp:346-348) and advance counters (:503-504)
    if constexpr (!FB) {
        if (deferred && lane == 0 && zlist) {
            const int slot = atomicAdd(c.zflag, 1);
            if (slot < zbatch) zlist[slot] = gi;            // (an instance appends once per step and the count starts at 0: always true)
        }
    }
    if (rollout_frame >= 0 && lane == 0 && !deferred && !(status & ISMPC_ST_Z_FAILED)) {   // a failed vertical solve is flagged, never fed back
        ismpc_tick_in* st = state_rw + gi;
        st->com_pos[0] = o_x; st->com_pos[1] = o_y; st->com_pos[2] = o_z;
        st->com_vel[0] = o_xd; st->com_vel[1] = o_yd; st->com_vel[2] = o_zd;
        st->simulation_time = w.sim;
        const int ctl = w.ctl + 1;
        st->control_iter = ctl;
        st->mpc_iter = (int)floor(ctl * c.cdt / c.dt);     // as written at Controller.cpp:504: 29*0.01/0.01 floors to 28, and parity keeps that
        st->footstep_counter = w.fc;
    }
}


// 8 workgroups (one wavefront per SIMD each) must be co-resident per CU: <= 64 VGPRs and -- the binding one on
// gfx950 -- <= 80 SGPRs (MI355X_MICROARCH.md "Residency": floor(800 / (ceil(sgpr/16)*16 + 16)) blocks per CU)
#ifndef ISMPC_AFF_WAVES
#define ISMPC_AFF_WAVES 4
#endif
// SW: a parameter-sweep handle at a horizon the lane-group kernels do not cover (128 < N <= 256): one instance per wavefront, so the
// instance's parameter set is wave-uniform and the body runs on that set's own record (tables and scalars), as the fallback launch does.
template <int R, bool SW = false>
__global__ __launch_bounds__(64 * ISMPC_AFF_WAVES) __attribute__((amdgpu_num_sgpr(80)))
void ismpc_tick_affine(const DevConst c, const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                       ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj, int batch, int rollout_frame,
                       unsigned char* zmark, int launch_id)
{
    const int lane = threadIdx.x & 63;
    const int gi = blockIdx.x * ISMPC_AFF_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (gi >= batch) return;
    if constexpr (SW) {
        const int ps = __builtin_amdgcn_readfirstlane((((rollout_frame >= 0) ? state_rw : in_ro) + gi)->reserved);
        const bool known = ps >= 0 && ps < c.nsets;                  // an unknown set: ISMPC_ST_BAD_INDEX, state passed through
        tick_affine_body<R, false>(c.sets[known ? ps : 0], gi, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id,
                                   zmark ? zlist_of(zmark, batch) : nullptr, batch, nullptr, known ? 0 : ISMPC_ST_BAD_INDEX);
    } else
    tick_affine_body<R, false>(c, gi, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, zmark ? zlist_of(zmark, batch) : nullptr, batch);
}

// =====================================================================================================================
// SEVERAL instances per wavefront, one group of LPI lanes each (horizons N <= 128): LPI = 16 (a DPP row, four instances per
// wavefront) or LPI = 8 (half a row, eight instances).  A horizon of 100 samples fills only 100 of the 128 sample slots of a
// wavefront and, worse, every scan, reduction and scalar of the tick is paid once per wavefront: with one instance per lane
// group the R = ceil(N/LPI) samples a lane owns are independent work for the FP64 pipe, the scans / reductions are log2(LPI)
// DPP steps inside a group (no cross-row fold, no readlane), and what used to be wave-uniform is group-uniform.  Same
// arithmetic as tick_affine_body; instances whose vertical QP has active inequality rows are deferred exactly as there.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp64b(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
// DPP move that writes every lane (rotations; shifts with bound_ctrl): no "old" operand, so no register to pre-load
template <int CTRL, bool BOUND_ZERO>
__device__ __forceinline__ double dpp64n(double src)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(src), CTRL, 0xf, 0xf, BOUND_ZERO);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(src), CTRL, 0xf, 0xf, BOUND_ZERO);
    return __hiloint2double(hi, lo);
}
// Lane-group primitives.  LPI = 16: the group is a DPP row.  LPI = 8: two groups per row; a shift by 4 is confined to its
// half row with the bank mask (banks 1 and 3 keep the zero / the old value), shifts by 1 and 2 zero the lanes whose source
// sits in the neighbouring group; sums are xor butterflies (quad_perm, quad_perm, row_half_mirror).
template <int LPI> struct Grp;
template <> struct Grp<16> {
    static constexpr int STEPS = 4;
    __device__ static __forceinline__ double sum(double v)             // sum over the group, in every lane (row_ror:1,2,4,8)
    {
        v += dpp64n<0x121, false>(v); v += dpp64n<0x122, false>(v); v += dpp64n<0x124, false>(v); v += dpp64n<0x128, false>(v);
        return v;
    }
    __device__ static __forceinline__ int sum_i(int v)
    {
        v += __builtin_amdgcn_mov_dpp(v, 0x121, 0xf, 0xf, false); v += __builtin_amdgcn_mov_dpp(v, 0x122, 0xf, 0xf, false);
        v += __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, false); v += __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false);
        return v;
    }
    // v from lane + 2^K of the group, 0 past its end
    template <int K> __device__ static __forceinline__ double shl0(double v, int) { return dpp64n<0x100 + (1 << K), true>(v); }
    // lane 0 of the group, in every lane: quad_perm [0,0,0,0], then row_shr:4 into bank 1, row_shr:8 into banks 2,3
    __device__ static __forceinline__ double bcast0(double v)
    {
        v = dpp64b<0x000, 0xf>(v, v); v = dpp64b<0x114, 0x2>(v, v); v = dpp64b<0x118, 0xc>(v, v);
        return v;
    }
    // v from the next lane of the group; the last lane gets `fill`
    __device__ static __forceinline__ double next_or(double fill, double v, int) { return dpp64<0x101, 0xf, false>(fill, v); }
};
// LPI = 32: two rows per group (two instances per wavefront; R = 4 samples per lane at N <= 128).  Row-local DPP steps as for
// 16, one exchange with the partner row (lane ^ 16: ds_swizzle) for the sums, readlane of lanes 16 / 48 for the scan's fifth step.
__device__ __forceinline__ double swz16(double v)       // the value of lane ^ 16
{
    return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F), __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F));
}
template <> struct Grp<32> {
    static constexpr int STEPS = 5;
    __device__ static __forceinline__ double sum(double v)
    {
        v += dpp64n<0x121, false>(v); v += dpp64n<0x122, false>(v); v += dpp64n<0x124, false>(v); v += dpp64n<0x128, false>(v);
        return v + swz16(v);
    }
    __device__ static __forceinline__ int sum_i(int v)
    {
        v += __builtin_amdgcn_mov_dpp(v, 0x121, 0xf, 0xf, false); v += __builtin_amdgcn_mov_dpp(v, 0x122, 0xf, 0xf, false);
        v += __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, false); v += __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false);
        return v + __builtin_amdgcn_ds_swizzle(v, 0x401F);
    }
    // steps 0-3 stay inside a row (v from lane + 2^K of the ROW, 0 past its end); step 4 is grp_scan_step's own
    template <int K> __device__ static __forceinline__ double shl0(double v, int) { return dpp64n<0x100 + (1 << K), true>(v); }
    __device__ static __forceinline__ double bcast0(double v)
    {
        const double a = readlane64<0>(v), b = readlane64<32>(v);
        return ((threadIdx.x & 32) != 0) ? b : a;
    }
    // v from the next lane of the group (wave_shl:1 crosses the row boundary); the last lane gets `fill`
    __device__ static __forceinline__ double next_or(double fill, double v, int li)
    {
        const double t = dpp64<0x130, 0xf, false>(fill, v);
        return (li == 31) ? fill : t;
    }
};
template <> struct Grp<8> {
    static constexpr int STEPS = 3;
    __device__ static __forceinline__ double sum(double v)
    {
        v += dpp64n<0x0B1, false>(v);                                  // quad_perm [1,0,3,2]
        v += dpp64n<0x04E, false>(v);                                  // quad_perm [2,3,0,1]
        v += dpp64n<0x141, false>(v);                                  // row_half_mirror: lane l <-> 7 - l of its half row
        return v;
    }
    __device__ static __forceinline__ int sum_i(int v)
    {
        v += __builtin_amdgcn_mov_dpp(v, 0x0B1, 0xf, 0xf, false); v += __builtin_amdgcn_mov_dpp(v, 0x04E, 0xf, 0xf, false);
        v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, false);
        return v;
    }
    template <int K> __device__ static __forceinline__ double shl0(double v, int li)
    {
        if constexpr (K == 2) return dpp64b<0x104, 0x5>(0.0, v);      // banks 0 and 2 read lanes + 4; banks 1 and 3 stay 0
        else { const double t = dpp64n<0x100 + (1 << K), true>(v); return (li + (1 << K) < 8) ? t : 0.0; }
    }
    __device__ static __forceinline__ double bcast0(double v)
    {
        v = dpp64b<0x000, 0xf>(v, v); v = dpp64b<0x114, 0xa>(v, v);    // quad_perm [0,0,0,0]; banks 1,3 <- banks 0,2
        return v;
    }
    __device__ static __forceinline__ double next_or(double fill, double v, int li)
    {
        const double t = dpp64<0x101, 0xf, false>(fill, v);
        return (li == 7) ? fill : t;
    }
};
// y <- T y for T = I + Tm taken from lane + 2^K of the group (Tm = 0 past the end of the group)
template <int LPI, int K>
__device__ __forceinline__ void grp_scan_step(M2& y, int li)
{
    if constexpr (LPI == 32 && K == 4) {
        // the rows have their own suffix products; the lower row of a group still needs the upper row's total (its lane 16 / 48)
        const bool g1 = (threadIdx.x & 32) != 0, low = li < 16;
        const double ya = g1 ? readlane64<48>(y.a) : readlane64<16>(y.a), yb = g1 ? readlane64<48>(y.b) : readlane64<16>(y.b);
        const double yc = g1 ? readlane64<48>(y.c) : readlane64<16>(y.c), yd = g1 ? readlane64<48>(y.d) : readlane64<16>(y.d);
        const double ta = low ? ya - 1.0 : 0.0, tb = low ? yb : 0.0, tc = low ? yc : 0.0, td = low ? yd - 1.0 : 0.0;
        M2 r;
        r.a = fma(ta, y.a, fma(tb, y.c, y.a)); r.b = fma(ta, y.b, fma(tb, y.d, y.b));
        r.c = fma(tc, y.a, fma(td, y.c, y.c)); r.d = fma(tc, y.b, fma(td, y.d, y.d));
        y = r;
    } else
    if constexpr (K < Grp<LPI>::STEPS) {
        const double ta = Grp<LPI>::template shl0<K>(y.a - 1.0, li), tb = Grp<LPI>::template shl0<K>(y.b, li);
        const double tc = Grp<LPI>::template shl0<K>(y.c, li), td = Grp<LPI>::template shl0<K>(y.d - 1.0, li);
        M2 r;
        r.a = fma(ta, y.a, fma(tb, y.c, y.a)); r.b = fma(ta, y.b, fma(tb, y.d, y.b));
        r.c = fma(tc, y.a, fma(td, y.c, y.c)); r.d = fma(tc, y.b, fma(td, y.d, y.d));
        y = r;
    }
}

// -DISMPC_STAMPS (diagnostic build, scripts/stamps_b.py): wall-clock stamps (s_memrealtime, 100 MHz) of every wavefront of the
// per-tick lane-group kernels at a few points of the tick; written to a buffer nothing else reads.
#ifdef ISMPC_STAMPS
__device__ unsigned long long g_stamps[16384 * 8];
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define STAMP(k_) do { if (g_stamp_wave >= 0 && g_stamp_wave < 16384) { const unsigned long long t_ = stamp_now(); if ((threadIdx.x & 63) == 0) g_stamps[g_stamp_wave * 8 + (k_)] = t_; } } while (0)
#define STAMP_DECL const int g_stamp_wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)
#else
#define STAMP(k_) do {} while (0)
#define STAMP_DECL do {} while (0)
#endif

// What one instance carries from tick to tick (group-uniform: every lane of the group holds the same values) and what a tick
// produces (valid in lane 0 of the group).
struct QState { double x, y, z, xd, yd, zd; Walk w; int ps; };     // ps: parameter set of the instance (sweep handles; -1 = invalid record)
struct QOut { double x, y, z, xd, yd, zd, uz0, ux0, uy0; int status, itx, ity; };

// One tick of one instance per lane group, registers in, registers out.  `s.w` is the WalkState the tick runs with (caller
// bookkeeping already applied).  Returns true in every lane of a group whose instance has active vertical inequality rows
// (deferred to the active-set fallback; its QOut is then provisional).
// LDS of one wavefront of the lane-group kernels: the midpoint window of each of its instances, staged so that the global
// loads are coalesced (lane li reads sample k LPI + li) and every lane then picks up its own R consecutive samples.  A lane's
// block starts at li * MIDM double2; MIDM is odd, which keeps the 16-byte reads of 16 lanes on 16 different bank quads.
template <int R> constexpr int midm() { return R | 1; }
template <int R, int LPI> constexpr int wave_lds_double2() { return (64 / LPI) * LPI * midm<R>(); }
// ... and, in the kernels that run the inequality fallback themselves, at least the fallback's working window (z_active_set)
template <int R, int LPI> constexpr int wave_lds_double2_fb() { return wave_lds_double2<R, LPI>() > (Z_LDS_DOUBLES + 1) / 2 ? wave_lds_double2<R, LPI>() : (Z_LDS_DOUBLES + 1) / 2; }

// KF: how the knapsack Newton loop is scheduled, not what it computes (the iterates are bit-identical): 0 = count the saturated
// samples first and form the two sums only for axes that still move (fewest instructions: batches that fill the chip are
// VALU-issue bound); 1 = count and sums of both axes in one pass, six interleaved group reductions instead of up to three
// dependent ones per axis.  Measured (scripts/kf_sweep.sh, MI355X): 1 is slower at every batch size -- 10.2 vs 9.7 us at 1 024
// instances, 14.0 vs 12.8 at 8 192, 51.8 vs 46.7 at 65 536, 5.9 vs 5.4 us per tick in the rollout kernel -- even one wavefront
// alone on its SIMD is bound by the number of instructions it issues, not by the reduction chains.  Kept as a build-time knob.
#ifndef ISMPC_KF_INLINE
#define ISMPC_KF_INLINE 0
#endif
#ifndef ISMPC_KF_MAIN
#define ISMPC_KF_MAIN 0
#endif
#ifndef ISMPC_KF_ROLLOUT
#define ISMPC_KF_ROLLOUT 0
#endif
// SW: parameter sweep -- the groups of a wavefront may belong to different parameter sets: what depends on the set (tables
// of the vertical stage, tails, mass, eta, box widths, bounds on S u) is read through the instance's own record c.sets[s.ps]
// (per-lane loads); horizon, plan, dt, g and the gate are the handle's.  SW = false compiles to exactly the plain kernel.
template <int R, int LPI, int KF, bool SW = false>
__device__ __forceinline__ bool tick_group_core(const DevConst& c, const int lane, const QState& s, QOut& o, double* __restrict__ u_traj_inst,
                                                double2* __restrict__ lds_wave)
{
    constexpr int NT = ismpc::Tables::NT;
    const int N = c.N;
    const int li = lane & (LPI - 1);                  // lane inside the group = inside the instance
    const double dt = c.dt;
    const Walk& w = s.w;
    const double x0 = s.x, y0 = s.y, z0 = s.z, xd0 = s.xd, yd0 = s.yd, zd0 = s.zd;
    const DevConst* P = SW ? c.sets + (s.ps >= 0 ? s.ps : 0) : nullptr;
    const double* p_vqT = SW ? P->vqT : c.vqT;
    const double p_z_lo = SW ? P->z_lo : c.z_lo, p_z_hi = SW ? P->z_hi : c.z_hi;
    const double p_inv_mass = SW ? P->inv_mass : c.inv_mass, p_inv_eta = SW ? P->inv_eta : c.inv_eta;
    const double p_half_run = SW ? P->half_run : c.half_run, p_half_first = SW ? P->half_first : c.half_first;
    const double* p_tailx = SW ? P->tailx : c.tailx; const double* p_taily = SW ? P->taily : c.taily;
    const double p_dt_over_mass = SW ? P->dt_over_mass : c.dt_over_mass, p_h_des = SW ? P->h_des : c.h_des;
    int idx;
    const int gate_status = gate_tick(c, w, idx) | ((SW && s.ps < 0) ? ISMPC_ST_BAD_INDEX : 0);     // group-uniform; a gated group runs the arithmetic on idx = 0 and drops it
    int status = gate_status;
    const bool run = gate_status == 0;
    if (!run) idx = 0;
    const int n0 = li * R;                            // this lane owns samples n0 .. n0+R-1 (tables are zero past N)
    STAMP_DECL;
    STAMP(1);                                         // the record has arrived (gate_tick consumed it)

    // ---- vertical stage from the affine tables (MPCSolver.cpp:223-243, is_running :262-263)
    const int pat = (run && w.fc > 1 && w.mpc < c.npat) ? w.mpc : c.npat;
    const double2* T = reinterpret_cast<const double2*>(p_vqT) + (size_t)pat * (R * 3 * LPI) + li;  // 3 x 16 bytes per sample, lane-contiguous
    // midpoint window [idx, idx + LPI R) of this instance -> LDS, coalesced (consumed after the scan; MPCSolver.cpp:328-338,388-389)
    constexpr int MIDM = midm<R>();
    double2* Lm = lds_wave + (lane / LPI) * (LPI * MIDM);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int j = k * LPI + li;
        Lm[(j / R) * MIDM + (j % R)] = reinterpret_cast<const double2*>(c.midxy)[min(idx + j, c.nmid - 1)];
    }
    double u[R], su[R];
    double smin = INFINITY, smax = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double2 t01 = T[(3 * r) * LPI], t23 = T[(3 * r + 1) * LPI], t45 = T[(3 * r + 2) * LPI];
        u[r] = fma(zd0, t23.x, fma(z0, t01.y, t01.x));
        su[r] = fma(zd0, t45.y, fma(z0, t45.x, t23.y));
        if (n0 + r < N) { smin = fmin(smin, su[r]); smax = fmax(smax, su[r]); }
    }
    if (!c.flat) {                                      // plans with mid_z != 0 (MPCSolver.cpp:259): per-frame offsets, pattern corrections
        int elo = 0, ne = 0;
        if (pat < c.npat) { elo = c.e_lo[pat]; ne = c.ne[pat]; }
        const int pp = pat < c.npat ? pat : 0;
        int nemax = 0;
#pragma unroll
        for (int g = 0; g < 64; g += LPI) nemax = max(nemax, __builtin_amdgcn_readlane(ne, g));
        const double* dUr = (SW ? P->dU : c.dU) + (size_t)idx * NT;          // (a sweep: this instance's set has its own offsets and corrections)
        const double* sUr = (SW ? P->SdU : c.SdU) + (size_t)idx * NT;
        double du[R], ds[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { du[r] = dUr[n0 + r]; ds[r] = sUr[n0 + r]; }
        for (int e = 0; e < nemax; ++e) {
            const bool on = e < ne;
            const double ue = on ? dUr[elo + e] : 0.0;
            const double* wr = (SW ? P->Wt : c.Wt) + ((size_t)pp * c.Fmax + (on ? e : 0)) * NT + n0;
            const double* sr = (SW ? P->SW : c.SW) + ((size_t)pp * c.Fmax + (on ? e : 0)) * NT + n0;
#pragma unroll
            for (int r = 0; r < R; ++r) { du[r] = fma(-wr[r], ue, du[r]); ds[r] = fma(-sr[r], ue, ds[r]); }
        }
        smin = INFINITY; smax = -INFINITY;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r;
            u[r] += du[r]; su[r] += ds[r];
            if (n >= elo && n < elo + ne) u[r] = 0.0;
            if (n < N) { smin = fmin(smin, su[r]); smax = fmax(smax, su[r]); }
        }
    }
    const double zlo_t = p_z_lo - 1e-11 * fmax(1.0, fabs(p_z_lo)), zhi_t = p_z_hi + 1e-11 * fmax(1.0, fabs(p_z_hi));
    const bool viol = smin < zlo_t || smax > zhi_t;                                                // MPCSolver.cpp:158-160, beyond rounding
    const unsigned long long vmask = __builtin_amdgcn_ballot_w64(viol);
    const bool deferred = run && (((vmask >> (lane & (64 - LPI))) & ((1ull << LPI) - 1ull)) != 0ull);
    if (deferred) status |= ISMPC_ST_Z_INEQ_ACTIVE;

    // ---- lambda_j (MPCSolver.cpp:306) and A_j, B_j (:353-361): A = [1+wQ, dt P; lam dt P, 1+wQ], B = [-wQ, -lam dt P]
    double ch1[R], s1[R], s2[R], lam0_l = 0.0;
    bool big = false, mid = false;
    {
        double wv_[R], le_[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 tq = reinterpret_cast<const double2*>(c.tzgT)[r * LPI + li];
            const double zpos = su[r] + fma(tq.x, zd0, z0) + tq.y;                  // S u + T_bar_z s + T_bar_g_z
            const double zacc = fma(p_inv_mass, u[r], -c.g);
            const double lam = (c.g + zacc) * frcp(zpos);
            if (r == 0) lam0_l = lam;
            le_[r] = (lam < c.gate) ? 0.0 : lam;
            const double dtn = (n0 + r < N) ? dt : 0.0;
            wv_[r] = le_[r] * dtn * dtn;
            s1[r] = dtn;                                  // dt_n for now
            big = big || (wv_[r] > 0.25);
            mid = mid || (wv_[r] > 0.004);
        }
        if (__builtin_amdgcn_ballot_w64(mid) == 0) {      // degree 3 is exact to < 1 ulp for w <= 0.004 (see tick_affine_body)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double wv = wv_[r];
                double P = 1.0 / 5040.0, Q = 1.0 / 40320.0;
                P = fma(P, wv, 1.0 / 120.0);         Q = fma(Q, wv, 1.0 / 720.0);
                P = fma(P, wv, 1.0 / 6.0);           Q = fma(Q, wv, 1.0 / 24.0);
                P = fma(P, wv, 1.0);                 Q = fma(Q, wv, 0.5);
                ch1[r] = wv * Q; s1[r] = s1[r] * P; s2[r] = le_[r] * s1[r];
            }
        } else {
            // some group of this wavefront needs the long polynomial.  The choice is made PER GROUP (= per instance): a group whose own
            // samples all have w <= 0.004 takes the degree-3 values here too, so an instance's record does not depend on which instances
            // share its wavefront (round 4: a sweep sorted by parameter set, ismpc_sweep_bind, changes an instance's wave-mates)
            const bool gmid = ((__builtin_amdgcn_ballot_w64(mid) >> (lane & (64 - LPI))) & ((LPI == 64) ? ~0ull : ((1ull << LPI) - 1ull))) != 0ull;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double wv = wv_[r], dtn = s1[r];
                double P = 1.0 / 1307674368000.0, Q = 1.0 / 20922789888000.0;
                P = fma(P, wv, 1.0 / 6227020800.0);  Q = fma(Q, wv, 1.0 / 87178291200.0);
                P = fma(P, wv, 1.0 / 39916800.0);    Q = fma(Q, wv, 1.0 / 479001600.0);
                P = fma(P, wv, 1.0 / 362880.0);      Q = fma(Q, wv, 1.0 / 3628800.0);
                P = gmid ? fma(P, wv, 1.0 / 5040.0) : 1.0 / 5040.0;   Q = gmid ? fma(Q, wv, 1.0 / 40320.0) : 1.0 / 40320.0;   // (degree 3 starts here)
                P = fma(P, wv, 1.0 / 120.0);         Q = fma(Q, wv, 1.0 / 720.0);
                P = fma(P, wv, 1.0 / 6.0);           Q = fma(Q, wv, 1.0 / 24.0);
                P = fma(P, wv, 1.0);                 Q = fma(Q, wv, 0.5);
                ch1[r] = wv * Q; s1[r] = dtn * P; s2[r] = le_[r] * s1[r];
                if (__builtin_amdgcn_ballot_w64(big) != 0 && wv > 0.25) {     // lambda dt^2 > 1/4: off any physical gait; libm
                    const double x = sqrt(wv);
                    ch1[r] = cosh(x) - 1.0; s1[r] = dtn * (sinh(x) / x); s2[r] = le_[r] * s1[r];
                }
            }
        }
    }
    // ---- inclusive suffix product over the group: Y_l = A(block LPI-1) ... A(block l); C_sc = [1, 1/eta]
    M2 Y = (M2){1.0 + ch1[0], s1[0], s2[0], 1.0 + ch1[0]};
#pragma unroll
    for (int r = 1; r < R; ++r) Y = mul((M2){1.0 + ch1[r], s1[r], s2[r], 1.0 + ch1[r]}, Y);
    STAMP(2);                                         // tables arrived, lambda / A_j / local products done
    grp_scan_step<LPI, 0>(Y, li); grp_scan_step<LPI, 1>(Y, li); grp_scan_step<LPI, 2>(Y, li); grp_scan_step<LPI, 3>(Y, li);
    grp_scan_step<LPI, 4>(Y, li);
    const double ie = p_inv_eta;
    const double cva = fma(ie, Y.c, Y.a), cvb = fma(ie, Y.d, Y.b);       // C_sc (suffix product from this lane's first sample)
    double c0 = Grp<LPI>::next_or(1.0, cva, li), c1 = Grp<LPI>::next_or(ie, cvb, li);                 // the lane needs it one lane up
    // ---- Aeq(n) = C_sc phi_input(:,n) = c_n B_n, walking the lane's samples backwards
    double a[R];
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
        a[r] = -fma(c0, ch1[r], c1 * s2[r]);
        const double k0 = fma(c0, ch1[r], fma(c1, s2[r], c0)), k1 = fma(c1, ch1[r], fma(c0, s1[r], c1));
        c0 = k0; c1 = k1;
    }
    const double h = (w.fc > 1) ? p_half_run : p_half_first;                                          // MPCSolver.cpp:328-338
    double q0 = 0.0, s_ax = 0.0, s_ay = 0.0, mx0 = 0.0, my0 = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();          // the staged window is complete
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = n0 + r;
        const double2 mq = Lm[li * MIDM + r];
        const double mx = (n < N) ? mq.x : 0.0, my = (n < N) ? mq.y : 0.0;
        if (r == 0) { mx0 = mx; my0 = my; }
        q0 = fma(a[r], a[r], q0); s_ax = fma(a[r], mx, s_ax); s_ay = fma(a[r], my, s_ay);
    }
    q0 = Grp<LPI>::sum(q0); s_ax = Grp<LPI>::sum(s_ax); s_ay = Grp<LPI>::sum(s_ay);
    // C_sc phi_state sits in lane 0 of the group (cva, cvb there); beq - a'mid (MPCSolver.cpp:381-384), group-uniform
    const double bpx = Grp<LPI>::bcast0((p_tailx[idx] - fma(cva, x0, cvb * xd0)) - s_ax);
    const double bpy = Grp<LPI>::bcast0((p_taily[idx] - fma(cva, y0, cvb * yd0)) - s_ay);
    const double sgx = (bpx < 0.0) ? -1.0 : 1.0, sgy = (bpy < 0.0) ? -1.0 : 1.0;
    STAMP(3);                                         // scan, backward walk, midpoints, reductions done
    // min 1/2|v|^2, a'v = bp, |v| <= h  ->  v_n = sg sign(a_n) min(tau |a_n|, h): Newton on the concave piecewise-linear
    // G(tau) = sum |a_n| min(tau |a_n|, h) from tau = 0; the groups iterate in lockstep, each with its own state
    const double Tq[2] = { fabs(bpx), fabs(bpy) };
    const double iq0 = frcp(q0);
    double tau[2] = { Tq[0] * iq0, Tq[1] * iq0 };
    int its[2] = {1, 1}, prev[2] = {0, 0};
    bool live[2] = {true, true};
    int st3 = 0;
    if (!(q0 > 0.0)) {                                                       // no sample can move the ZMP
#pragma unroll
        for (int ax = 0; ax < 2; ++ax) {
            tau[ax] = (Tq[ax] > 0.0) ? INFINITY : 0.0;
            if (Tq[ax] > 1e-300) st3 |= (ax == 0 ? ISMPC_ST_X_INFEASIBLE : ISMPC_ST_Y_INFEASIBLE);
        }
    }
    if constexpr (KF == 0) {
        for (int it = 0; it < N + 2; ++it) {
            if (__builtin_amdgcn_ballot_w64(live[0] || live[1]) == 0ull) break;
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
                if (__builtin_amdgcn_ballot_w64(live[ax]) == 0ull) continue;      // this axis is done in every group of the wavefront (the other one
                                                                                  // keeps the loop alive for 0.6 more rounds on average: scripts/knapsack_hist.py)
                int cl = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) cl += (tau[ax] * fabs(a[r]) >= h) ? 1 : 0;
                const int cnt = Grp<LPI>::sum_i(cl);
                if (live[ax] && cnt == prev[ax]) live[ax] = false;                // active set unchanged: exact
                if (__builtin_amdgcn_ballot_w64(live[ax]) == 0ull) continue;
                double ssat = 0.0, qfree = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    // 0 / 1 masks and two fused multiply-adds instead of two 64-bit selects and two adds: the same sums bit for bit
                    // (fma(1, x, s) = s + x rounded once, fma(0, x, s) = s), a third fewer instructions in the loop the kernel spends most in
                    const double ab = fabs(a[r]);
                    const bool sat = tau[ax] * ab >= h;
                    const double ms = sat ? 1.0 : 0.0, mf = sat ? 0.0 : 1.0;
                    ssat = fma(ms, ab, ssat); qfree = fma(mf, a[r] * a[r], qfree);
                }
                ssat = Grp<LPI>::sum(ssat); qfree = Grp<LPI>::sum(qfree);
                if (live[ax]) {
                    ++its[ax];
                    const double rem = fma(-h, ssat, Tq[ax]);
                    if (!(qfree > 0.0)) {                                         // everything saturated
                        if (rem > fma(h * ssat, 1e-12, 1e-300)) st3 |= (ax == 0 ? ISMPC_ST_X_INFEASIBLE : ISMPC_ST_Y_INFEASIBLE);
                        tau[ax] = INFINITY; live[ax] = false;
                    } else {
                        const double tn = rem * frcp(qfree);
                        if (!(tn > tau[ax])) live[ax] = false;
                        else { tau[ax] = tn; prev[ax] = cnt; }
                    }
                }
            }
        }
    } else {
        for (int it = 0; it < N + 2; ++it) {
            if (__builtin_amdgcn_ballot_w64(live[0] || live[1]) == 0ull) break;
            int cl[2] = {0, 0};
            double ssat[2] = {0.0, 0.0}, qfree[2] = {0.0, 0.0};
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool sat = tau[ax] * fabs(a[r]) >= h;
                    cl[ax] += sat ? 1 : 0; ssat[ax] += sat ? fabs(a[r]) : 0.0;
                    const double a2 = a[r] * a[r]; qfree[ax] += sat ? 0.0 : a2;
                }
            }
            const int cnt0 = Grp<LPI>::sum_i(cl[0]), cnt1 = Grp<LPI>::sum_i(cl[1]);
            ssat[0] = Grp<LPI>::sum(ssat[0]); ssat[1] = Grp<LPI>::sum(ssat[1]);
            qfree[0] = Grp<LPI>::sum(qfree[0]); qfree[1] = Grp<LPI>::sum(qfree[1]);
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
                const int cnt = ax == 0 ? cnt0 : cnt1;
                if (live[ax] && cnt == prev[ax]) live[ax] = false;            // active set unchanged: exact
                if (live[ax]) {
                    ++its[ax];
                    const double rem = fma(-h, ssat[ax], Tq[ax]);
                    if (!(qfree[ax] > 0.0)) {                                 // everything saturated
                        if (rem > fma(h * ssat[ax], 1e-12, 1e-300)) st3 |= (ax == 0 ? ISMPC_ST_X_INFEASIBLE : ISMPC_ST_Y_INFEASIBLE);
                        tau[ax] = INFINITY; live[ax] = false;
                    } else {
                        const double tn = rem * frcp(qfree[ax]);
                        if (!(tn > tau[ax])) live[ax] = false;
                        else { tau[ax] = tn; prev[ax] = cnt; }
                    }
                }
            }
        }
    }

    STAMP(4);                                         // knapsack Newton done
    // ---- lane 0 of the group finishes the instance: integration (MPCSolver.cpp:274-278, 406-422)
    o.x = x0; o.y = y0; o.z = z0; o.xd = xd0; o.yd = yd0; o.zd = zd0;
    o.uz0 = 0.0; o.ux0 = 0.0; o.uy0 = 0.0; o.itx = 0; o.ity = 0;
    if (li == 0 && run) {
        o.uz0 = u[0];
        o.z = fma(dt, zd0, z0);
        o.zd = fma(p_dt_over_mass, o.uz0, zd0) - dt * c.g;
        if (isnan(o.z)) { o.z = p_h_des; status |= ISMPC_ST_Z_NAN; }
        if (isnan(o.zd)) { o.zd = 0.0; status |= ISMPC_ST_Z_NAN; }
        const double A0a = 1.0 + ch1[0], A0b = s1[0], A0c = s2[0];
        if (lam0_l > c.gate) {                                            // MPCSolver.cpp:322
            status |= st3; o.itx = its[0]; o.ity = its[1];
            const double sa0 = (a[0] < 0.0) ? -1.0 : 1.0;
            o.ux0 = fma(sgx * sa0, (fabs(a[0]) > 0.0) ? fmin(tau[0] * fabs(a[0]), h) : 0.0, mx0);
            o.uy0 = fma(sgy * sa0, (fabs(a[0]) > 0.0) ? fmin(tau[1] * fabs(a[0]), h) : 0.0, my0);
        } else status |= ISMPC_ST_FLIGHT;
        o.x  = fma(1.0 - A0a, o.ux0, fma(A0a, x0, A0b * xd0));
        o.xd = fma(-A0c, o.ux0, fma(A0c, x0, A0a * xd0));
        o.y  = fma(1.0 - A0a, o.uy0, fma(A0a, y0, A0b * yd0));
        o.yd = fma(-A0c, o.uy0, fma(A0c, y0, A0a * yd0));
    }
    o.status = status;
    if (u_traj_inst) {
        const bool stage3 = run && Grp<LPI>::bcast0(lam0_l) > c.gate;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r;
            if (n < N) {
                double vx = 0.0, vy = 0.0;
                if (stage3) {
                    const double sa = (a[r] < 0.0) ? -1.0 : 1.0;
                    vx = fma(sgx * sa, (fabs(a[r]) > 0.0) ? fmin(tau[0] * fabs(a[r]), h) : 0.0, c.midx[idx + n]);
                    vy = fma(sgy * sa, (fabs(a[r]) > 0.0) ? fmin(tau[1] * fabs(a[r]), h) : 0.0, c.midy[idx + n]);
                }
                u_traj_inst[n] = run ? u[r] : 0.0; u_traj_inst[N + n] = vx; u_traj_inst[2 * N + n] = vy;
            }
        }
    }
    return deferred;
}

__device__ __forceinline__ void store_record(ismpc_tick_out* __restrict__ rec, const QOut& o)
{
    double2* o2 = reinterpret_cast<double2*>(rec);
    const long long packed = (long long)(unsigned)o.status | ((long long)(unsigned)((o.itx & 255) | ((o.ity & 255) << 8)) << 32);
    o2[0] = make_double2(o.x, o.y); o2[1] = make_double2(o.z, o.xd); o2[2] = make_double2(o.yd, o.zd);
    o2[3] = make_double2(o.uz0, o.ux0); o2[4] = make_double2(o.uy0, __longlong_as_double(packed));
}
// Controller.cpp:346-348 (feed the output back), :503-504 (advance the counters)
__device__ __forceinline__ void store_feedback(const DevConst& c, ismpc_tick_in* __restrict__ st, const QOut& o, const Walk& w)
{
    st->com_pos[0] = o.x; st->com_pos[1] = o.y; st->com_pos[2] = o.z;
    st->com_vel[0] = o.xd; st->com_vel[1] = o.yd; st->com_vel[2] = o.zd;
    st->simulation_time = w.sim;
    const int ctl = w.ctl + 1;
    st->control_iter = ctl;
    st->mpc_iter = (int)floor(ctl * c.cdt / c.dt);     // as written at Controller.cpp:504 (see tick_affine_body)
    st->footstep_counter = w.fc;
}

// One launch = one tick: record in, record out (and, in the host-driven closed loop, state fed back in place)
template <int R, int LPI, int KF, bool SW = false>
__device__ __forceinline__ bool tick_group_body(const DevConst& c, const int gi_raw, const int batch, const int lane,
                                                const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                                                ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj,
                                                int rollout_frame, unsigned char* zmark, int launch_id, double2* __restrict__ lds_wave, int* zlist = nullptr)
{
    const bool valid = gi_raw < batch;
    const int gi = valid ? gi_raw : batch - 1;        // tail groups recompute the last instance and store nothing
    STAMP_DECL;
    STAMP(0);                                         // first instructions of the wavefront
    const ismpc_tick_in* rec = ((rollout_frame >= 0) ? state_rw : in_ro) + gi;
    QState s;
    s.w = load_walk(c, rec, rollout_frame);
    s.x = rec->com_pos[0]; s.y = rec->com_pos[1]; s.z = rec->com_pos[2];
    s.xd = rec->com_vel[0]; s.yd = rec->com_vel[1]; s.zd = rec->com_vel[2];
    s.ps = 0;
    if (SW) { const int ps = rec->reserved; s.ps = (ps >= 0 && ps < c.nsets) ? ps : -1; }       // an unknown set: ISMPC_ST_BAD_INDEX, state passed through
    QOut o;
    const bool deferred = tick_group_core<R, LPI, KF, SW>(c, lane, s, o, (u_traj && valid) ? u_traj + (size_t)gi * 3 * c.N : nullptr, lds_wave);
    if ((lane & (LPI - 1)) == 0 && valid) {
        if (out) store_record(out + gi, o);
        if (deferred) {
            if (c.zseen) *c.zseen = launch_id;
            if (zlist) { const int slot = atomicAdd(c.zflag, 1); if (slot < batch) zlist[slot] = gi; }
        }
        if (rollout_frame >= 0 && !deferred) store_feedback(c, state_rw + gi, o, s.w);
    }
    STAMP(5);                                         // stores issued
    return deferred && valid;
}

#ifndef ISMPC_QUAD_WAVES
#define ISMPC_QUAD_WAVES 4
#endif
// Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b mod 8, each with its own L2).  The virtual block of
// workgroup b: XCD x takes the contiguous range [x q + min(x, r), ...) of the nb blocks (q = nb / 8, r = nb mod 8) -- a bijection.
__device__ __forceinline__ int sweep_vblock(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    return x * q + min(x, r) + (b >> 3);
}
// instance of launch slot `slot` (see DevConst::order); slots past the batch name no instance
template <bool SW> __device__ __forceinline__ int slot_instance(const DevConst& c, int slot, int batch)
{
    if (SW) { if (c.order) return slot < batch ? c.order[slot] : batch; }
    return slot;
}
template <int R, int LPI, bool SW = false>
__global__ __launch_bounds__(64 * ISMPC_QUAD_WAVES)
void ismpc_tick_quad(const DevConst c, const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                     ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj, int batch, int rollout_frame,
                     unsigned char* zmark, int launch_id)
{
    constexpr int IPW = 64 / LPI;                      // instances per wavefront
    __shared__ double2 lds_mid[ISMPC_QUAD_WAVES][wave_lds_double2<R, LPI>()];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = (SW && c.order) ? sweep_vblock(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int wave = blk * ISMPC_QUAD_WAVES + wv;
    if (wave * IPW >= batch) return;
    tick_group_body<R, LPI, ISMPC_KF_MAIN, SW>(c, slot_instance<SW>(c, wave * IPW + lane / LPI, batch), batch, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, lds_mid[wv],
                                               zmark ? zlist_of(zmark, batch) : nullptr);
}

// The inequality fallback as a real CALL from the one-launch kernel: inlined there, its 200 registers' worth of state made the
// hot path of every tick spill 180 scalar registers; called, the tick keeps the register allocation of ismpc_tick_quad and only a
// wavefront that does defer an instance pays for the call.
template <int RW>
__device__ __attribute__((noinline)) void fallback_call(const DevConst* cp, int gi, int lane, const ismpc_tick_in* in_ro, ismpc_tick_in* state_rw,
                                                        ismpc_tick_out* out, double* u_traj, int rollout_frame, unsigned char* zmark, int launch_id, double* zlds)
{
    tick_affine_body<RW, true>(*cp, gi, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, nullptr, 0, zlds);
}

// Latency variant for small batches (every wavefront resident at once): a wavefront that deferred one of its
// instances runs the inequality fallback for it right away, with all 64 lanes, so a step is ONE launch.
template <int R, int LPI, int RW>
__global__ __launch_bounds__(64 * ISMPC_QUAD_WAVES, 2)      // two wavefronts per SIMD (that is all a batch that takes this kernel has)
void ismpc_tick_quad_inline(const DevConst c, const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                            ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj, int batch, int rollout_frame,
                            unsigned char* zmark, int launch_id, const DevConst* __restrict__ cdev)
{
    constexpr int IPW = 64 / LPI;
    __shared__ double2 lds_mid[ISMPC_QUAD_WAVES][wave_lds_double2_fb<R, LPI>()];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = blockIdx.x * ISMPC_QUAD_WAVES + wv;
    if (wave * IPW >= batch) return;
    const bool def = tick_group_body<R, LPI, ISMPC_KF_INLINE>(c, wave * IPW + lane / LPI, batch, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, lds_mid[wv]);
    unsigned long long m = __builtin_amdgcn_ballot_w64(def);
    if (m == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int q = 0; q < IPW; ++q)
        if ((m >> (LPI * q)) & 1ull)
            fallback_call<RW>(cdev, wave * IPW + q, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, reinterpret_cast<double*>(lds_mid[wv]));   // (the constants in memory:
                                                                                     // taking the address of the by-value argument would move the hot path's copy to the stack)
}

// The one-launch form for batches that do NOT fit the chip at once: the tick keeps the three wavefronts per SIMD of ismpc_tick_quad
// (the kernel asks for them, and the compiler hands that register budget down to the fallback it calls: the fallback spills to
// scratch instead, and only a wavefront that defers an instance runs it).  No second, normally idle, launch per step: +1-2 % at
// 65 536 instances, +4 % at 32 768, +8 % at 16 384 (same box, scripts/ab_env.sh ISMPC_ONE_LAUNCH=0).  SW: parameter sweeps, the
// fallback runs on the deferred instance's own set.
// wavefronts per SIMD of ismpc_tick_quad<R, LPI, SW> (profiles/r03/kernel_resources.md): what the one-launch form asks for
#ifndef ISMPC_OCC_R13
#define ISMPC_OCC_R13 2
#endif
template <int R, bool SW> constexpr int one_occ() { return R <= 4 ? (SW ? 3 : 4) : R <= 7 ? 3 : R == 8 ? (SW ? 2 : 3) : R <= 13 ? ISMPC_OCC_R13 : 1; }
template <int RW, int OCC>        // OCC: one copy per residency target (the register budget comes down from the calling kernels)
__device__ __attribute__((noinline))
void fallback_call_one(const DevConst* cp, int gi, int lane, const ismpc_tick_in* in_ro, ismpc_tick_in* state_rw,
                       ismpc_tick_out* out, double* u_traj, int rollout_frame, unsigned char* zmark, int launch_id, double* zlds)
{
    tick_affine_body<RW, true>(*cp, gi, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, nullptr, 0, zlds);
}
template <int R, int LPI, int RW, bool SW>
__global__ __launch_bounds__(64 * ISMPC_QUAD_WAVES, (one_occ<R, SW>()))
void ismpc_tick_quad_one(const DevConst c, const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                         ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj, int batch, int rollout_frame,
                         unsigned char* zmark, int launch_id, const DevConst* __restrict__ cdev)
{
    constexpr int IPW = 64 / LPI;
    __shared__ double2 lds_mid[ISMPC_QUAD_WAVES][wave_lds_double2_fb<R, LPI>()];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = (SW && c.order) ? sweep_vblock(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int wave = blk * ISMPC_QUAD_WAVES + wv;
    if (wave * IPW >= batch) return;
    const bool def = tick_group_body<R, LPI, ISMPC_KF_MAIN, SW>(c, slot_instance<SW>(c, wave * IPW + lane / LPI, batch), batch, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, lds_mid[wv]);
    unsigned long long m = __builtin_amdgcn_ballot_w64(def);
    if (m == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int q = 0; q < IPW; ++q)
        if ((m >> (LPI * q)) & 1ull) {
            const int gi = __builtin_amdgcn_readfirstlane(slot_instance<SW>(c, wave * IPW + q, batch));
            const DevConst* cp = cdev;
            if (SW) cp = c.sets + __builtin_amdgcn_readfirstlane((((rollout_frame >= 0) ? state_rw : in_ro) + gi)->reserved);   // (a deferred instance has a valid set)
            fallback_call_one<RW, one_occ<R, SW>()>(cp, gi, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, reinterpret_cast<double*>(lds_mid[wv]));
        }
}

// Closed loop inside ONE launch (Controller.cpp:297-310 bookkeeping, :346-348 feedback, :503-504 counters): instances are
// independent, so a wavefront keeps the state of its instances in registers for `ticks` ticks and writes one trajectory
// record per tick; nothing but the read-only tables is re-read.  Bit-identical to `ticks` launches of the per-tick kernels
// (same tick_group_core, same fallback body).
//   FB = false (the rollout itself): an instance whose vertical inequality rows become active at tick t parks its pre-tick
//     state in `state`, records t in stop_tick and sits out the rest of the launch;
//   FB = true (second launch, exits at once unless the first one parked something): one wavefront per parked instance
//     resumes it at its tick, running the active-set fallback (all 64 lanes, through memory) at the ticks that need it.
// Keeping the fallback out of the first kernel keeps its register budget that of the tick itself.
template <int R, int LPI, int RW, bool FB, bool SW = false>
__global__ __launch_bounds__(64 * ISMPC_QUAD_WAVES, 2)
void ismpc_rollout_quad(const DevConst c, ismpc_tick_in* state, ismpc_tick_out* __restrict__ traj, int batch, int first_frame, int ticks,
                        int* __restrict__ stop_tick, int launch_id)
{
    constexpr int IPW = 64 / LPI;
    __shared__ double2 lds_mid[ISMPC_QUAD_WAVES][FB ? wave_lds_double2_fb<R, LPI>() : wave_lds_double2<R, LPI>()];
    const int lane = threadIdx.x & 63, li = lane & (LPI - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = blockIdx.x * ISMPC_QUAD_WAVES + wv;
    if constexpr (FB) { if (*(volatile int*)(c.zflag + 2) == 0) return; }      // nothing parked (workgroup-uniform: the first launch is done)
    const int nwork = FB ? batch : (batch + IPW - 1) / IPW;          // FB: one instance per wavefront (every group computes it, group 0 stores)
    for (int work = wave; work < nwork; work += FB ? (int)gridDim.x * ISMPC_QUAD_WAVES : nwork) {
        const int gi_raw = FB ? work : work * IPW + lane / LPI;
        const bool valid = FB ? (lane < LPI) : (gi_raw < batch);
        const int gi = (gi_raw < batch) ? gi_raw : batch - 1;
        int t0 = 0;
        if constexpr (FB) { t0 = stop_tick[gi]; if (t0 < 0) continue; }
        ismpc_tick_in* rec = state + gi;
        QState s;
        s.w.sim = rec->simulation_time; s.w.mpc = rec->mpc_iter; s.w.ctl = rec->control_iter; s.w.fc = rec->footstep_counter;
        s.x = rec->com_pos[0]; s.y = rec->com_pos[1]; s.z = rec->com_pos[2];
        s.xd = rec->com_vel[0]; s.yd = rec->com_vel[1]; s.zd = rec->com_vel[2]; s.ps = 0;
        if (SW) { const int ps = rec->reserved; s.ps = (ps >= 0 && ps < c.nsets) ? ps : -1; }     // sweep handles: the instance's parameter set
        bool alive = true;                                              // FB = false: false once the instance is parked
        int stopped = -1;
        for (int t = t0; t < ticks; ++t) {
            const int frame = first_frame + t;
            // caller bookkeeping in front of solve(): Controller.cpp:297-304 (enabled) and :310 -- load_walk's rollout branch
            const Walk before = s.w;
            if (s.w.fc >= 0 && s.w.fc < c.rows && s.w.sim >= c.ftsp_t[s.w.fc] - 1) { s.w.ctl = 0; s.w.mpc = 0; s.w.fc = s.w.fc + 1; }
            s.w.sim = (double)frame;
            QOut o;
            const bool def = tick_group_core<R, LPI, ISMPC_KF_ROLLOUT, SW>(c, lane, s, o, nullptr, lds_mid[wv]);
            const bool park = def && alive;
            if (li == 0 && valid && alive && !def && traj) store_record(traj + (size_t)t * batch + gi, o);
            // a deferred instance: its pre-tick state goes to memory (FB = false: to stay there; FB = true: for the fallback body)
            if (park && valid && li == 0) {
                rec->com_pos[0] = s.x; rec->com_pos[1] = s.y; rec->com_pos[2] = s.z;
                rec->com_vel[0] = s.xd; rec->com_vel[1] = s.yd; rec->com_vel[2] = s.zd;
                rec->simulation_time = before.sim; rec->mpc_iter = before.mpc; rec->control_iter = before.ctl; rec->footstep_counter = before.fc;
            }
            if constexpr (!FB) {
                if (park) { alive = false; stopped = t; }
            }
            // feedback (Controller.cpp:346-348) and counters (:503-504), in registers; lane 0 of the group holds the result
            s.x = Grp<LPI>::bcast0(o.x); s.y = Grp<LPI>::bcast0(o.y); s.z = Grp<LPI>::bcast0(o.z);
            s.xd = Grp<LPI>::bcast0(o.xd); s.yd = Grp<LPI>::bcast0(o.yd); s.zd = Grp<LPI>::bcast0(o.zd);
            s.w.ctl = s.w.ctl + 1;
            s.w.mpc = (int)floor(s.w.ctl * c.cdt / c.dt);
            if constexpr (FB) {
                if (__builtin_amdgcn_ballot_w64(def && valid) != 0ull) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    // (FB: one instance per wavefront, its set is wave-uniform and valid -- an invalid one never defers)
                    tick_affine_body<RW, true>(SW ? c.sets[__builtin_amdgcn_readfirstlane(max(s.ps, 0))] : c, gi, lane, nullptr, state,
                                               traj ? traj + (size_t)t * batch : nullptr, nullptr, frame, nullptr, 0, nullptr, 0, reinterpret_cast<double*>(lds_mid[wv]));
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    const volatile ismpc_tick_in* vr = rec;
                    s.x = vr->com_pos[0]; s.y = vr->com_pos[1]; s.z = vr->com_pos[2];
                    s.xd = vr->com_vel[0]; s.yd = vr->com_vel[1]; s.zd = vr->com_vel[2];
                    s.w.sim = vr->simulation_time; s.w.mpc = vr->mpc_iter; s.w.ctl = vr->control_iter; s.w.fc = vr->footstep_counter;
                }
            }
        }
        if (li == 0 && valid) {
            if (alive) {
                rec->com_pos[0] = s.x; rec->com_pos[1] = s.y; rec->com_pos[2] = s.z;
                rec->com_vel[0] = s.xd; rec->com_vel[1] = s.yd; rec->com_vel[2] = s.zd;
                rec->simulation_time = s.w.sim; rec->mpc_iter = s.w.mpc; rec->control_iter = s.w.ctl; rec->footstep_counter = s.w.fc;
            }
            if constexpr (!FB) {
                stop_tick[gi] = stopped;
                if (stopped >= 0) atomicAdd(c.zflag + 2, 1);
            }
        }
    }
    if constexpr (FB) {      // the last resume workgroup zeroes the parked count for the next rollout (see DevConst::zflag)
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(c.zflag + 3, 1) == (int)gridDim.x - 1) { c.zflag[2] = 0; c.zflag[3] = 0; __threadfence(); }
    }
}

// Second launch of every tick of a large batch: exits at once unless the first one deferred instances (active inequality rows).
template <int R, bool SW = false>
__global__ __launch_bounds__(256)
void ismpc_tick_affine_fallback(const DevConst c, const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                                ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj, int batch, int rollout_frame,
                                unsigned char* zmark, int launch_id)
{
    const int* zl = zlist_of(zmark, batch);
    const int ndef = min(*(volatile int*)c.zflag, batch);     // stable while this launch runs (the appending kernel is done): workgroup-uniform
    if (ndef == 0) return;
    __shared__ double zlds[4][Z_LDS_DOUBLES];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave0 = blockIdx.x * 4 + wv;
    double* const zwin = zlds[wv];
    for (int k = wave0; k < ndef; k += gridDim.x * 4) {
        const int gi = __builtin_amdgcn_readfirstlane(zl[k]);
        {
            if (SW) {
                // one instance per wavefront: its parameter set is wave-uniform, the body runs on that set's own record
                const int ps = __builtin_amdgcn_readfirstlane((((rollout_frame >= 0) ? state_rw : in_ro) + gi)->reserved);
                tick_affine_body<R, true>(c.sets[ps], gi, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, nullptr, 0, zwin);    // (a deferred instance has a valid set)
            } else tick_affine_body<R, true>(c, gi, lane, in_ro, state_rw, out, u_traj, rollout_frame, zmark, launch_id, nullptr, 0, zwin);
        }
    }
    // every workgroup has read the count by the time it gets here; the last one to arrive hands the counters back zeroed
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(c.zflag + 1, 1) == (int)gridDim.x - 1) { c.zflag[0] = 0; c.zflag[1] = 0; __threadfence(); }
}

// ---- ismpc_sweep_bind: counting sort of the instances of a batch by parameter set (bucket nsets: records that name no set) ----------
__global__ __launch_bounds__(256) void sweep_sort_hist(const ismpc_tick_in* __restrict__ in, int batch, int nsets, int* __restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const int ps = in[i].reserved;
    atomicAdd(counts + ((ps >= 0 && ps < nsets) ? ps : nsets), 1);
}
// exclusive scan of counts[0 .. n) in place (one workgroup; n <= 65 536 + 1): counts[k] becomes the first slot of bucket k
__global__ __launch_bounds__(256) void sweep_sort_scan(int* __restrict__ counts, int n)
{
    __shared__ int part[256];
    const int tid = threadIdx.x, per = (n + 255) / 256, lo = tid * per, hi = min(lo + per, n);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += counts[k];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int k = 0; k < 256; ++k) { const int v = part[k]; part[k] = run; run += v; } }
    __syncthreads();
    int run = part[tid];
    for (int k = lo; k < hi; ++k) { const int v = counts[k]; counts[k] = run; run += v; }
}
__global__ __launch_bounds__(256) void sweep_sort_scatter(const ismpc_tick_in* __restrict__ in, int batch, int nsets, int* __restrict__ cursor, int* __restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const int ps = in[i].reserved;
    order[atomicAdd(cursor + ((ps >= 0 && ps < nsets) ? ps : nsets), 1)] = i;      // (the order INSIDE a bucket is whatever the atomics give: no result depends on it)
}

// ------------------------------------------------------------------------
// Entry points run on the handle's device and leave the caller's current device as they found it (a torch process that
// drives several GPUs keeps allocating where it was).
struct DeviceGuard {
    int prev = -1, dev; hipError_t err = hipSuccess;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
thread_local std::string g_err = "";
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(ISMPC_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define ON_DEVICE(h_) DeviceGuard guard_((h_)->device); HIP_TRY(guard_.err)

}  // namespace

struct ismpc_handle {
    ismpc::Tables t;
    DevConst c{};
    int device = 0;
    std::vector<void*> dev_allocs;
    // staging for the host-pointer entry point
    ismpc_tick_in* st_in = nullptr; ismpc_tick_out* st_out = nullptr; int st_cap = 0;
    ismpc_tick_in* pin_in = nullptr; ismpc_tick_out* pin_out = nullptr;   // host-mapped staging for small batches (PIN_BATCH records)
    bool pin_off = false;
    hipStream_t own_stream = nullptr;
    int host_mode = 3;                                          // ISMPC_HOST_MODE: bit 0 = kernel reads page-locked caller records in place, bit 1 = writes them in place
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false; bool timed_pending = false; double last_ms = 0.0;
    int* zseen_host = nullptr; // DevConst::zseen as the host sees it
    int one_launch = 2;       // 2: one launch per step (ismpc_tick_quad_inline up to the resident size; beyond it ismpc_tick_quad_one unless recent launches deferred instances); 1: only the former; 0: never; 3: always
    int force_waves = 0;      // dense path: 4, 8 or 16 wavefronts per workgroup (0 = 16)
    unsigned char* zmark = nullptr; int zmark_cap = 0; int launch_id = 0; bool z_fallback = true;
    int* zstop = nullptr; int zstop_cap = 0;     // in-kernel rollouts: tick at which an instance was handed to the resume launch (-1: never)
    bool dense_path = false;  // true: per-tick MFMA solve (ismpc_tick_dense); false: affine tables (ismpc_tick_affine)
    int cus = 0;              // compute units of the device (kernel variant selection); 0: never the one-launch variant
    bool quad_path = true;    // affine tables, several instances per wavefront (ismpc_tick_quad) where it applies; ISMPC_PATH=wave: one per wavefront
    int lpi = 16;             // lanes per instance of the quad kernels: 16 (four instances per wavefront), 8 (eight) or 32 (two); ISMPC_LPI
    bool lpi_auto = true;     // no ISMPC_LPI: 32 lanes per instance for batches of <= LPI32_BATCH instances (scripts/lpi_batch.py: 1 us of 9-10
                              // there, slower from 3 072 on), 16 otherwise; the tables exist in both layouts
    const double* vqT32 = nullptr; const double* tzgT32 = nullptr;
    const double* vqT8 = nullptr; const double* tzgT8 = nullptr;     // ... and 8 lanes per instance beyond LPI16_BATCH instances per launch
    const DevConst* sets8 = nullptr;                                  // sweep handles: the set records with the 8-lane tables (null: 16 lanes at every batch size)
    bool kernel_rollout = true;   // closed loops run inside one launch (ismpc_rollout_quad); ISMPC_ROLLOUT=host: one launch per tick
    DevConst* c_dev = nullptr;    // the constants in device memory (the one-launch kernel's fallback call reads them there)
    bool sweep = false;           // ismpc_create_sweep: K parameter sets, tables built on the device (csrc/ismpc_sweep.hip)
    ismpc::SweepSlabs sw; std::vector<ismpc_params> sets; std::vector<double> ftsp;   // (the plan as given: ismpc_sweep_verify_tables rebuilds a set on the host)
    int* order = nullptr; int order_cap = 0, order_batch = 0;   // ismpc_sweep_bind: instances of the bound batch sorted by parameter set (+ nsets + 1 bucket cursors)
    hipStream_t last_stream = nullptr; bool used = false;   // stream of the previous launch: zmark / zstop outlive a call and are re-allocated
                                                            // only after that stream has drained (grow_sync)
};

namespace {

template <typename T>
int upload(ismpc_handle* h, const std::vector<T>& v, const T** dst)
{
    void* p = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    h->dev_allocs.push_back(p);
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return ISMPC_OK;
}

// Shape of the lane-group kernels for horizon N: samples per lane R (the smallest instantiated value that covers N) for
// LPI = 16 / 8 lanes per instance; RW = samples per lane of the one-instance-per-wavefront fallback body.
constexpr int LPI32_BATCH = 2048;
// ... 8 lanes per instance (eight instances per wavefront, 13 samples per lane at N = 100, two wavefronts per SIMD) beyond this many:
// a launch that does not fit the chip at once is bound by the instructions it issues, and a group reduction or scan step serves twice
// the instances (round 3, one box: 65 536 instances 1.30-1.35 -> 1.41-1.44e9 ticks/s, 32 768 1.18 -> 1.24e9, 16 384 0.94 -> 1.03e9;
// 8 192 level, 1 024 slower)
constexpr int LPI16_BATCH = 8192;
struct LaneLayout { int lpi; const double* vqT; const double* tzgT; };
LaneLayout pick_layout(const ismpc_handle* h, int batch, bool per_tick);
int quad_R(int N, int lpi)
{
    const int need = (N + lpi - 1) / lpi;
    if (lpi == 32) return 4;
    if (lpi == 16) return need <= 4 ? 4 : (need <= 7 ? 7 : 8);
    return need <= 8 ? 8 : (need <= 13 ? 13 : 16);
}
#define ISMPC_SHAPES(X) \
    if (lpi == 32) { if (h->c.N <= 64) X(4, 32, 1); else X(4, 32, 2); } else \
    if (lpi == 16) { if (RQ == 4) X(4, 16, 1); else if (RQ == 7) X(7, 16, 2); else X(8, 16, 2); } \
    else           { if (RQ == 8) X(8, 8, 1);   else if (RQ == 13) X(13, 8, 2); else X(16, 8, 2); }

// Scratch that outlives the call that allocated it (zmark, zstop) is used by later calls on whatever stream those pass: before
// it is re-allocated on stream `s`, the previous launch's stream -- if it is another one -- is drained.
// Lanes per instance of a launch of `batch` instances (the handle's own layout unless it chooses per launch: no ISMPC_LPI, no sweep).
// per_tick = false: the closed loop (ismpc_rollout_device, in the kernel or one launch per tick) keeps the 16-lane shape beyond LPI32_BATCH --
// with the state in registers across ticks it is the faster one there too (65 536 instances: 1.72 against 1.61e9 ticks/s).  Layouts differ in
// summation order, i.e. in the last bits: a tick of ismpc_solve_batch* and a tick of a rollout agree to rounding, not to the byte, beyond
// LPI16_BATCH instances per launch.
LaneLayout pick_layout(const ismpc_handle* h, int batch, bool per_tick)
{
    if (h->lpi_auto && h->vqT32 && batch <= LPI32_BATCH) return {32, h->vqT32, h->tzgT32};
    if ((h->lpi_auto || (h->sweep && h->sets8)) && h->vqT8 && per_tick && batch > LPI16_BATCH) return {8, h->vqT8, h->tzgT8};
    return {h->lpi, h->c.vqT, h->c.tzgT};
}
hipError_t grow_sync(ismpc_handle* h, hipStream_t s)
{
    if (h->used && h->last_stream != s) return hipStreamSynchronize(h->last_stream);
    return hipSuccess;
}
// Page-locked AND device-mapped over its whole length: both ends of [p, p + bytes) are host allocations known to the runtime and the
// device addresses of the two ends are `bytes - 1` apart (one mapping, or adjacent ones that continue each other).  A registration that
// covers only the head of the buffer, or an interior pointer near the end of a pinned block, fails this and takes the staged path
// instead of letting the kernel touch unmapped host memory over PCIe.
bool host_is_pinned(const void* p, size_t bytes)
{
    if (!p || bytes == 0) return false;
    hipPointerAttribute_t a, b;
    const char* last = static_cast<const char*>(p) + (bytes - 1);
    if (hipPointerGetAttributes(&a, p) != hipSuccess || hipPointerGetAttributes(&b, last) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost || b.type != hipMemoryTypeHost) return false;
    return a.devicePointer && b.devicePointer && static_cast<const char*>(b.devicePointer) - static_cast<const char*>(a.devicePointer) == (ptrdiff_t)(bytes - 1);
}
struct StreamMark { ismpc_handle* h; hipStream_t s; ~StreamMark() { h->last_stream = s; h->used = true; } };

int launch(ismpc_handle* h, int batch, const ismpc_tick_in* in, ismpc_tick_in* state, ismpc_tick_out* out,
           double* u_traj, int rollout_frame, hipStream_t s)
{
    if (batch <= 0) return ISMPC_OK;
    StreamMark mark_{h, s};
    const int R = (h->c.N + 63) / 64;
    if (!h->dense_path) {
        // fast path: wavefront per instance, 4 per workgroup; then the (normally empty) inequality fallback
        const dim3 grid((batch + 3) / 4), block(256);
        if (h->z_fallback && batch > h->zmark_cap) {
            // stream-ordered growth (no device-wide synchronisation inside an asynchronous entry point); callers that
            // capture graphs size it beforehand with ismpc_reserve
            HIP_TRY(grow_sync(h, s));
            if (h->zmark) HIP_TRY(hipFreeAsync(h->zmark, s));
            h->zmark = nullptr; h->zmark_cap = 0;
            HIP_TRY(hipMallocAsync((void**)&h->zmark, zscratch_bytes(batch), s));
            h->zmark_cap = batch;
        }
        unsigned char* zm = h->z_fallback ? h->zmark : nullptr;
        const int lid = ++h->launch_id;
        const dim3 fgrid(std::min((batch + 3) / 4, 256));      // the fallback walks the list of deferred instances: one wavefront each
        // Batches beyond the resident size: ONE launch (ismpc_tick_quad_one: a wavefront that defers an instance runs the fallback for it
        // itself) while nothing is being deferred -- the usual case, and the second launch would be idle; TWO launches (the deferred
        // list, one wavefront per deferred instance at the fallback's own register budget) when one of the last few launches did defer:
        // measured on the sweep batch (0.4 % deferred) 70.5 against 75.8 us per step.  The kernels leave the id of a deferring launch in
        // a word of host memory; reading it here costs nothing.  The host enqueues far ahead of the device (a closed loop of per-tick
        // launches is hundreds of launches deep), so the word is stale by that much: `recent` is the last 4 096 launches.  Both forms
        // give the same bytes, so a late switch costs microseconds, never correctness.
        const bool recent_deferrals = h->zseen_host && *(volatile int*)h->zseen_host != 0 && lid - *(volatile int*)h->zseen_host <= 4096;
        const bool one_big = zm && (h->one_launch == 3 || (h->one_launch == 2 && !recent_deferrals));
        // default for N <= 128: several instances per wavefront (ismpc_tick_quad); ISMPC_PATH=wave keeps one per wavefront
        if (h->quad_path && h->c.N <= 128) {
            const LaneLayout lay = pick_layout(h, batch, rollout_frame < 0);      // (a closed loop driven from the host keeps the in-kernel loop's layout: same bytes)
            const int lpi = lay.lpi, RQ = quad_R(h->c.N, lpi);
            DevConst cq = h->c;
            cq.vqT = lay.vqT; cq.tzgT = lay.tzgT;
            const int waves = (batch * lpi + 63) / 64;
            const dim3 qgrid((waves + ISMPC_QUAD_WAVES - 1) / ISMPC_QUAD_WAVES), qblock(64 * ISMPC_QUAD_WAVES);
            if (h->sweep) {
                // parameter sweep: the per-tick kernel reads each instance's set through c.sets (16 lanes per instance, 8 beyond LPI16_BATCH)
                if (lpi == 8) cq.sets = h->sets8;
                cq.order = (h->order && h->order_batch == batch && rollout_frame < 0) ? h->order : nullptr;     // (ismpc_sweep_bind; per-tick launches only)
#define ISMPC_SWEEP_SHAPES(X) \
    if (lpi == 16) { if (RQ == 4) X(4, 16, 1); else if (RQ == 7) X(7, 16, 2); else X(8, 16, 2); } \
    else           { if (RQ == 8) X(8, 8, 1);   else if (RQ == 13) X(13, 8, 2); else X(16, 8, 2); }
                if (one_big) {
#define ISMPC_QUADS1(RR, LL, RW_) hipLaunchKernelGGL((ismpc_tick_quad_one<RR, LL, RW_, true>), qgrid, qblock, 0, s, cq, in, state, out, u_traj, batch, rollout_frame, zm, lid, (const DevConst*)h->c_dev)
                    ISMPC_SWEEP_SHAPES(ISMPC_QUADS1)
#undef ISMPC_QUADS1
                    HIP_TRY(hipGetLastError());
                    return ISMPC_OK;
                }
#define ISMPC_QUADS(RR, LL, RW_) hipLaunchKernelGGL((ismpc_tick_quad<RR, LL, true>), qgrid, qblock, 0, s, cq, in, state, out, u_traj, batch, rollout_frame, zm, lid)
                ISMPC_SWEEP_SHAPES(ISMPC_QUADS)
#undef ISMPC_QUADS
#undef ISMPC_SWEEP_SHAPES
                if (zm) {
                    if (R == 1) hipLaunchKernelGGL((ismpc_tick_affine_fallback<1, true>), fgrid, block, 0, s, h->c, in, state, out, u_traj, batch, rollout_frame, zm, lid);
                    else        hipLaunchKernelGGL((ismpc_tick_affine_fallback<2, true>), fgrid, block, 0, s, h->c, in, state, out, u_traj, batch, rollout_frame, zm, lid);
                }
                HIP_TRY(hipGetLastError());
                return ISMPC_OK;
            }
            // every wavefront resident at once (<= 2 per SIMD) and a fallback to run: one launch that handles deferred instances itself
            if (zm && h->one_launch >= 1 && h->cus > 0 && waves <= 8 * h->cus) {
#define ISMPC_QUADI(RR, LL, RW_) hipLaunchKernelGGL((ismpc_tick_quad_inline<RR, LL, RW_>), qgrid, qblock, 0, s, cq, in, state, out, u_traj, batch, rollout_frame, zm, lid, (const DevConst*)h->c_dev)
                ISMPC_SHAPES(ISMPC_QUADI)
#undef ISMPC_QUADI
                HIP_TRY(hipGetLastError());
                return ISMPC_OK;
            }
            if (one_big) {                  // any other batch size: one launch too, at the tick's own three wavefronts per SIMD
#define ISMPC_QUADB(RR, LL, RW_) hipLaunchKernelGGL((ismpc_tick_quad_one<RR, LL, RW_, false>), qgrid, qblock, 0, s, cq, in, state, out, u_traj, batch, rollout_frame, zm, lid, (const DevConst*)h->c_dev)
                ISMPC_SHAPES(ISMPC_QUADB)
#undef ISMPC_QUADB
                HIP_TRY(hipGetLastError());
                return ISMPC_OK;
            }
#define ISMPC_QUAD(RR, LL, RW_) hipLaunchKernelGGL((ismpc_tick_quad<RR, LL>), qgrid, qblock, 0, s, cq, in, state, out, u_traj, batch, rollout_frame, zm, lid)
            ISMPC_SHAPES(ISMPC_QUAD)
#undef ISMPC_QUAD
            if (zm) {
                if (R == 1) hipLaunchKernelGGL(ismpc_tick_affine_fallback<1>, fgrid, block, 0, s, h->c, in, state, out, u_traj, batch, rollout_frame, zm, lid);
                else        hipLaunchKernelGGL(ismpc_tick_affine_fallback<2>, fgrid, block, 0, s, h->c, in, state, out, u_traj, batch, rollout_frame, zm, lid);
            }
            HIP_TRY(hipGetLastError());
            return ISMPC_OK;
        }
#define ISMPC_AFF(RR) do { \
        if (h->sweep) { \
            hipLaunchKernelGGL((ismpc_tick_affine<RR, true>), dim3((batch + ISMPC_AFF_WAVES - 1) / ISMPC_AFF_WAVES), dim3(64 * ISMPC_AFF_WAVES), 0, s, h->c, in, state, out, u_traj, batch, rollout_frame, zm, lid); \
            if (zm) hipLaunchKernelGGL((ismpc_tick_affine_fallback<RR, true>), fgrid, block, 0, s, h->c, in, state, out, u_traj, batch, rollout_frame, zm, lid); \
        } else { \
        hipLaunchKernelGGL(ismpc_tick_affine<RR>, dim3((batch + ISMPC_AFF_WAVES - 1) / ISMPC_AFF_WAVES), dim3(64 * ISMPC_AFF_WAVES), 0, s, h->c, in, state, out, u_traj, batch, rollout_frame, zm, lid); \
        if (zm) hipLaunchKernelGGL(ismpc_tick_affine_fallback<RR>, fgrid, block, 0, s, h->c, in, state, out, u_traj, batch, rollout_frame, zm, lid); \
        } \
    } while (0)
        switch (R) {
            case 1: ISMPC_AFF(1); break;
            case 2: ISMPC_AFF(2); break;
            case 3: ISMPC_AFF(3); break;
            case 4: ISMPC_AFF(4); break;
            default: return fail(ISMPC_E_UNSUPPORTED, "horizon N > 256");
        }
#undef ISMPC_AFF
        HIP_TRY(hipGetLastError());
        return ISMPC_OK;
    }
    // dense path (kept for A/B and as the per-tick MFMA formulation): 16 instances per workgroup
    const dim3 grid((batch + TI - 1) / TI);
    const size_t lds = (size_t)TI * h->c.NPs * sizeof(double);
    const int waves = h->force_waves ? h->force_waves : 16;
#define ISMPC_LAUNCH(RR, WW) hipLaunchKernelGGL((ismpc_tick_dense<RR, WW>), grid, dim3(64 * WW), lds, s, h->c, in, state, out, u_traj, batch, rollout_frame)
#define ISMPC_LAUNCH_R(RR) do { if (waves == 16) ISMPC_LAUNCH(RR, 16); else if (waves == 8) ISMPC_LAUNCH(RR, 8); else ISMPC_LAUNCH(RR, 4); } while (0)
    switch (R) {
        case 1: ISMPC_LAUNCH_R(1); break;
        case 2: ISMPC_LAUNCH_R(2); break;
        case 3: ISMPC_LAUNCH_R(3); break;
        case 4: ISMPC_LAUNCH_R(4); break;
        default: return fail(ISMPC_E_UNSUPPORTED, "horizon N > 256");
    }
#undef ISMPC_LAUNCH_R
#undef ISMPC_LAUNCH
    HIP_TRY(hipGetLastError());
    return ISMPC_OK;
}

}  // namespace

extern "C" {

int ismpc_abi_version(void) { return ISMPC_ABI_VERSION; }
const char* ismpc_last_error(void) { return g_err.c_str(); }

void ismpc_params_default(ismpc_params* p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->mpc_dt = 0.01; p->control_dt = 0.01;                 // parameters.cpp:9-10
    const double ss = 0.35, ds = 0.1, pred = 1.0;           // parameters.cpp:11-13
    p->N = (int)std::lround(pred / p->mpc_dt);              // :42
    p->S = (int)std::lround(ss / p->mpc_dt);                // :43
    p->F = (int)std::lround(ds / p->mpc_dt);                // :44
    p->M = 2;                                               // :45
    p->mass = 50.0; p->g = 9.81; p->h_des = 0.69;           // :39,40,16
    p->foot_width = 0.09; p->first_step_halfwidth = 1.0;    // :21 ; MPCSolver.cpp:334-337
    p->q_p = 1005000.0; p->q_u = 0.01; p->q_v = 100.0;      // MPCSolver.cpp:253-255
    p->z_ineq_lo = 0.0; p->z_ineq_hi = 10000.0;             // MPCSolver.cpp:159-160
    p->lambda_gate = 2.0;                                   // MPCSolver.cpp:322
}

static int create_impl(const ismpc_params* params, int K, bool sweep, const double* ftsp, int rows, int device, ismpc_handle** out)
{
    if (!params || !ftsp || !out) return fail(ISMPC_E_INVALID, "null argument");
    *out = nullptr;
    if (sweep) {
        // the sets of a sweep share what fixes the shape of the problem; everything else may differ from set to set
        if (K < 1 || K > 65535) return fail(ISMPC_E_INVALID, "a sweep holds 1 .. 65535 parameter sets");
        for (int k = 0; k < K; ++k) {
            const ismpc_params& a = params[0]; const ismpc_params& b = params[k];
            if (a.N != b.N || a.S != b.S || a.F != b.F || a.M != b.M || a.mpc_dt != b.mpc_dt || a.control_dt != b.control_dt || a.g != b.g || a.lambda_gate != b.lambda_gate)
                return fail(ISMPC_E_INVALID, "the parameter sets of a sweep share N, S, F, M, mpc_dt, control_dt, g and lambda_gate");
            if (!(b.mass > 0) || !(b.h_des > 0) || !(b.q_u > 0) || b.q_p < 0 || b.q_v < 0) return fail(ISMPC_E_INVALID, "sweep: mass, h_des, q_u must be positive, q_p and q_v non-negative");
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(ISMPC_E_NO_DEVICE, "no HIP device visible: the ISMPC hot path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(ISMPC_E_INVALID, "device ordinal out of range");
    ismpc_handle* h = new (std::nothrow) ismpc_handle();
    if (!h) return fail(ISMPC_E_ALLOC, "out of host memory");
    std::string err;
    int rc = ismpc::build_tables(*params, ftsp, rows, h->t, err);
    if (rc != ISMPC_OK) { delete h; return fail(rc, err); }
    h->device = device;
    if (const char* fw = std::getenv("ISMPC_WAVES")) {            // tuning knob: wavefronts per workgroup
        const int v = std::atoi(fw);
        if (v == 4 || v == 8 || v == 16) h->force_waves = v;
    }
    if (const char* pth = std::getenv("ISMPC_PATH")) { h->dense_path = std::strcmp(pth, "dense") == 0; h->quad_path = std::strcmp(pth, "wave") != 0 && !h->dense_path; }
    if (const char* zf = std::getenv("ISMPC_Z_FALLBACK")) h->z_fallback = std::atoi(zf) != 0;   // 0: flag only, no second launch
    if (const char* lp = std::getenv("ISMPC_LPI")) { const int v = std::atoi(lp); if (v == 8 || v == 16 || v == 32) { h->lpi = v; h->lpi_auto = false; } }
    if (const char* ro = std::getenv("ISMPC_ROLLOUT")) h->kernel_rollout = std::strcmp(ro, "host") != 0;
    if (sweep) {      // one kernel shape: 16 lanes per instance (ISMPC_Z_FALLBACK=0 still means flag-only: bench.py times the tick kernel alone with it)
        h->sweep = true; h->lpi = 16; h->lpi_auto = false; h->quad_path = true; h->dense_path = false;
        h->sets.assign(params, params + K); h->ftsp.assign(ftsp, ftsp + (size_t)rows * 4);
    }
    DeviceGuard guard_(device);
    if (guard_.err != hipSuccess) { delete h; return fail(ISMPC_E_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err)); }
    { hipDeviceProp_t prop; h->cus = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 0; }
    if (const char* fu = std::getenv("ISMPC_ONE_LAUNCH")) h->one_launch = std::max(0, std::min(3, std::atoi(fu)));    // A/B: 0 = always two launches, 1 = one launch only for batches resident at once, 3 = always one    // 0: always two launches (A/B)
    const ismpc::Tables& t = h->t;
    DevConst& c = h->c;
    c.N = t.p.N; c.NP = t.NP; c.NPs = t.NP + 2; c.S = t.p.S; c.F = t.p.F; c.nmid = t.nmid; c.npat = t.npat;
    c.Fmax = t.Fmax; c.rows = t.rows; c.tick_divisor = t.tick_divisor;
    c.dt = t.p.mpc_dt; c.cdt = t.p.control_dt; c.mass = t.p.mass; c.g = t.p.g; c.h_des = t.p.h_des;
    c.half_run = t.p.foot_width / 2; c.half_first = t.p.first_step_halfwidth;
    c.q_p = t.p.q_p; c.q_u = t.p.q_u; c.q_v = t.p.q_v; c.z_lo = t.p.z_ineq_lo; c.z_hi = t.p.z_ineq_hi;
    c.gate = t.p.lambda_gate; c.eta = t.eta;
    c.inv_mass = 1.0 / t.p.mass; c.dt_over_mass = t.p.mpc_dt / t.p.mass; c.inv_eta = 1.0 / t.eta;
    c.sim_div = t.p.mpc_dt / t.p.control_dt; c.cdt_over_dt = t.p.control_dt / t.p.mpc_dt;
    rc = upload(h, t.Hinv, &c.Hinv);
    if (rc == ISMPC_OK) rc = upload(h, t.W, &c.W);
    if (rc == ISMPC_OK) rc = upload(h, t.midx, &c.midx);
    if (rc == ISMPC_OK) rc = upload(h, t.midy, &c.midy);
    if (rc == ISMPC_OK) rc = upload(h, t.midz, &c.midz);
    if (rc == ISMPC_OK) rc = upload(h, t.tailx, &c.tailx);
    if (rc == ISMPC_OK) rc = upload(h, t.taily, &c.taily);
    if (rc == ISMPC_OK) rc = upload(h, t.ftsp_t, &c.ftsp_t);
    if (rc == ISMPC_OK) rc = upload(h, t.e_lo, &c.e_lo);
    if (rc == ISMPC_OK) rc = upload(h, t.ne, &c.ne);
    if (rc == ISMPC_OK) rc = upload(h, t.vtab, &c.vtab);
    if (rc == ISMPC_OK) rc = upload(h, t.tz, &c.tz);
    if (rc == ISMPC_OK) rc = upload(h, t.tg, &c.tg);
    if (rc == ISMPC_OK) rc = upload(h, t.dU, &c.dU);
    if (rc == ISMPC_OK) rc = upload(h, t.SdU, &c.SdU);
    if (rc == ISMPC_OK) rc = upload(h, t.Wt, &c.Wt);
    if (rc == ISMPC_OK) rc = upload(h, t.SW, &c.SW);
    if (rc == ISMPC_OK) rc = upload(h, t.HSt, &c.HSt);
    if (rc == ISMPC_OK) rc = upload(h, t.SHSt, &c.SHSt);
    if (rc == ISMPC_OK) { std::vector<int> zf(4, 0); const int* zp = nullptr; rc = upload(h, zf, &zp); c.zflag = const_cast<int*>(zp); }
    if (rc == ISMPC_OK) {
        // one word of page-locked host memory the kernels write the launch id to when they defer an instance (see launch())
        void* hp = nullptr; void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            h->zseen_host = static_cast<int*>(hp); *h->zseen_host = 0; c.zseen = static_cast<int*>(dp);
        } else { if (hp) (void)hipHostFree(hp); (void)hipGetLastError(); c.zseen = nullptr; }
    }
    if (rc == ISMPC_OK) {
        // active-set fallback pool: 256 slots of (cap x cap + 4 cap + NT) doubles + cap ints, cap = N rows (every row may be active)
        c.zslots = 256; c.zcap = t.p.N; c.zldsq = Z_LDS_Q;
        if (const char* e = std::getenv("ISMPC_Z_LDS_Q")) c.zldsq = std::min(Z_LDS_Q, std::max(1, std::atoi(e)));
        c.zstride = (size_t)c.zcap * c.zcap + 4 * (size_t)c.zcap + ismpc::Tables::NT + ((size_t)c.zcap + 1) / 2 + 8;
        std::vector<int> busy(c.zslots, 0); const int* bp = nullptr;
        rc = upload(h, busy, &bp); c.zbusy = const_cast<int*>(bp);
        if (rc == ISMPC_OK) {
            void* zp = nullptr;
            if (hipMalloc(&zp, c.zstride * c.zslots * sizeof(double)) != hipSuccess) rc = fail(ISMPC_E_ALLOC, "fallback pool allocation failed");
            else { h->dev_allocs.push_back(zp); c.zpool = static_cast<double*>(zp); }
        }
    }
    if (rc == ISMPC_OK) {
        constexpr int NTq = ismpc::Tables::NT;
        const size_t npp = t.vtab.size() / (6 * (size_t)NTq);
        std::vector<double> vq(t.vtab.size()), tzg(2 * (size_t)NTq), mxy(2 * t.midx.size());
        for (size_t pp = 0; pp < npp; ++pp)
            for (int k = 0; k < 6; ++k)
                for (int n = 0; n < NTq; ++n) vq[(pp * NTq + n) * 6 + k] = t.vtab[(pp * 6 + k) * NTq + n];
        for (int n = 0; n < NTq; ++n) { tzg[2 * n] = t.tz[n]; tzg[2 * n + 1] = t.tg[n]; }
        for (size_t n = 0; n < t.midx.size(); ++n) { mxy[2 * n] = t.midx[n]; mxy[2 * n + 1] = t.midy[n]; }
        rc = upload(h, vq, &c.vq);
        if (rc == ISMPC_OK) rc = upload(h, tzg, &c.tzg);
        if (rc == ISMPC_OK) rc = upload(h, mxy, &c.midxy);
        if (rc == ISMPC_OK && t.p.N <= 128) {
            // lane-contiguous copies for the lane-group kernels' shapes (sample li*R + r of pattern p): the handle's layout and,
            // when the layout is chosen per launch, the 32-lane and the 8-lane one beside it
            for (int pass = 0; pass < ((h->lpi_auto || sweep) ? 3 : 1) && rc == ISMPC_OK; ++pass) {
                const int lpi = pass == 0 ? h->lpi : (pass == 1 ? 32 : 8), R = quad_R(t.p.N, lpi);
                std::vector<double> vqT(npp * (size_t)R * 3 * lpi * 2), tzgT((size_t)R * lpi * 2);
                for (size_t pp = 0; pp < npp; ++pp)
                    for (int r = 0; r < R; ++r)
                        for (int k = 0; k < 3; ++k)
                            for (int li = 0; li < lpi; ++li) {
                                const int n = li * R + r;                       // < 128 <= NT
                                const size_t dst = (((pp * R + r) * 3 + k) * lpi + li) * 2;
                                vqT[dst] = vq[(pp * NTq + n) * 6 + 2 * k]; vqT[dst + 1] = vq[(pp * NTq + n) * 6 + 2 * k + 1];
                            }
                for (int r = 0; r < R; ++r)
                    for (int li = 0; li < lpi; ++li) { const int n = li * R + r; tzgT[((size_t)r * lpi + li) * 2] = t.tz[n]; tzgT[((size_t)r * lpi + li) * 2 + 1] = t.tg[n]; }
                rc = upload(h, vqT, pass == 0 ? &c.vqT : (pass == 1 ? &h->vqT32 : &h->vqT8));
                if (rc == ISMPC_OK) rc = upload(h, tzgT, pass == 0 ? &c.tzgT : (pass == 1 ? &h->tzgT32 : &h->tzgT8));
            }
        }
    }
    c.flat = t.flat ? 1 : 0;
    if (rc != ISMPC_OK) { ismpc_destroy(h); return rc; }
    if (const char* hm = std::getenv("ISMPC_HOST_MODE")) h->host_mode = std::atoi(hm) & 3;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
        hipEventCreate(&h->ev1) != hipSuccess) { ismpc_destroy(h); return fail(ISMPC_E_NO_DEVICE, "stream/event creation failed"); }
    if (sweep) {
        // every set's tables, built on the device (MFMA Newton-Schulz inverse of the K vertical Hessians, csrc/ismpc_sweep.hip), and one
        // DevConst record per set: the handle's, with the set's scalars and table pointers in place of set 0's host-built ones
        std::string serr;
        // 8 lanes per instance beyond LPI16_BATCH, as plain handles take them (ISMPC_LPI=16 keeps 16 at every size).  With eight instances
        // of a wavefront reading eight sets' tables it measured level to 2 % slower on the 64-set batch (round 3: 9.25-9.28 against
        // 9.27-9.43e8 ticks/s); with the batch sorted by set (ismpc_sweep_bind: one set per wavefront) it is 5 % faster (round 4: 9.65-9.71
        // against 9.17-9.26e8), so it is the default now
        const char* lp8 = std::getenv("ISMPC_LPI");
        const bool lanes8 = !(lp8 && std::atoi(lp8) == 16);
        rc = ismpc::sweep_build(params, K, h->t, c.midx, c.midy, c.midz, c.e_lo, c.ne, 16, quad_R(t.p.N, 16), lanes8 ? 8 : 0, quad_R(t.p.N, 8), h->own_stream, h->sw, h->dev_allocs, serr);
        if (rc != ISMPC_OK) { ismpc_destroy(h); return fail(rc, serr); }
        std::vector<DevConst> cs((size_t)K, h->c);
        for (int k = 0; k < K; ++k) {
            DevConst& d = cs[k]; const ismpc_params& q = params[k];
            const double eta = std::sqrt(q.g / q.h_des);
            d.mass = q.mass; d.h_des = q.h_des; d.half_run = q.foot_width / 2; d.half_first = q.first_step_halfwidth;
            d.q_p = q.q_p; d.q_u = q.q_u; d.q_v = q.q_v; d.z_lo = q.z_ineq_lo; d.z_hi = q.z_ineq_hi; d.eta = eta;
            d.inv_mass = 1.0 / q.mass; d.dt_over_mass = q.mpc_dt / q.mass; d.inv_eta = 1.0 / eta;
            d.vtab = h->sw.vtab + (size_t)k * h->sw.s_vtab; d.vqT = h->sw.vqT + (size_t)k * h->sw.s_vqT;
            d.Wt = h->sw.Wt + (size_t)k * h->sw.s_W; d.SW = h->sw.SW + (size_t)k * h->sw.s_W;
            d.HSt = h->sw.HSt + (size_t)k * h->sw.s_HS; d.SHSt = h->sw.SHSt + (size_t)k * h->sw.s_HS;
            if (h->sw.dU) { d.dU = h->sw.dU + (size_t)k * h->sw.s_dU; d.SdU = h->sw.SdU + (size_t)k * h->sw.s_dU; }      // (plans with mid_z != 0)
            d.tailx = h->sw.tailx + (size_t)k * h->sw.s_tail; d.taily = h->sw.taily + (size_t)k * h->sw.s_tail;
            d.Hinv = nullptr; d.W = nullptr; d.vq = nullptr; d.sets = nullptr; d.nsets = 0;
        }
        void* sp = nullptr;
        if (hipMalloc(&sp, sizeof(DevConst) * (size_t)K) != hipSuccess) { ismpc_destroy(h); return fail(ISMPC_E_ALLOC, "sweep: set records allocation failed"); }
        h->dev_allocs.push_back(sp);
        if (hipMemcpy(sp, cs.data(), sizeof(DevConst) * (size_t)K, hipMemcpyHostToDevice) != hipSuccess) { ismpc_destroy(h); return fail(ISMPC_E_NO_DEVICE, "sweep: set records upload failed"); }
        h->c.sets = static_cast<const DevConst*>(sp); h->c.nsets = K;
        if (h->sw.vqT2) {                                      // the same records over the 8-lane copy of the affine tables (pick_layout)
            for (int k = 0; k < K; ++k) cs[k].vqT = h->sw.vqT2 + (size_t)k * h->sw.s_vqT2;
            void* sp8 = nullptr;
            if (hipMalloc(&sp8, sizeof(DevConst) * (size_t)K) != hipSuccess) { ismpc_destroy(h); return fail(ISMPC_E_ALLOC, "sweep: set records allocation failed"); }
            h->dev_allocs.push_back(sp8);
            if (hipMemcpy(sp8, cs.data(), sizeof(DevConst) * (size_t)K, hipMemcpyHostToDevice) != hipSuccess) { ismpc_destroy(h); return fail(ISMPC_E_NO_DEVICE, "sweep: set records upload failed"); }
            h->sets8 = static_cast<const DevConst*>(sp8);
        }
    }
    {
        void* cp = nullptr;
        if (hipMalloc(&cp, sizeof(DevConst)) != hipSuccess) { ismpc_destroy(h); return fail(ISMPC_E_ALLOC, "constants allocation failed"); }
        h->dev_allocs.push_back(cp); h->c_dev = static_cast<DevConst*>(cp);
        if (hipMemcpy(cp, &h->c, sizeof(DevConst), hipMemcpyHostToDevice) != hipSuccess) { ismpc_destroy(h); return fail(ISMPC_E_NO_DEVICE, "constants upload failed"); }
    }
    *out = h;
    return ISMPC_OK;
}

int ismpc_create(const ismpc_params* params, const double* ftsp, int rows, int device, ismpc_handle** out)
{
    return create_impl(params, 1, false, ftsp, rows, device, out);
}

int ismpc_create_sweep(const ismpc_params* params, int n_sets, const double* ftsp, int rows, int device, ismpc_handle** out)
{
    return create_impl(params, n_sets, true, ftsp, rows, device, out);
}

int ismpc_sweep_bind(ismpc_handle* h, int batch, const ismpc_tick_in* in_dev, void* stream)
{
    if (!h || batch < 0 || (batch > 0 && !in_dev)) return fail(ISMPC_E_INVALID, "bad argument");
    if (!h->sweep) return fail(ISMPC_E_INVALID, "ismpc_sweep_bind needs a handle of ismpc_create_sweep");
    ON_DEVICE(h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (batch == 0) { h->order_batch = 0; return ISMPC_OK; }
    const int nb = h->c.nsets + 1;
    if (batch + nb > h->order_cap) {
        HIP_TRY(hipDeviceSynchronize());                       // (a set-up call: launches that still read the old order finish first)
        if (h->order) HIP_TRY(hipFree(h->order));
        h->order = nullptr; h->order_cap = 0; h->order_batch = 0;
        HIP_TRY(hipMalloc((void**)&h->order, sizeof(int) * (size_t)(batch + nb)));
        h->order_cap = batch + nb;
    }
    int* cursor = h->order + batch;
    HIP_TRY(hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)nb, s));
    const dim3 grid((batch + 255) / 256), block(256);
    hipLaunchKernelGGL(sweep_sort_hist, grid, block, 0, s, in_dev, batch, h->c.nsets, cursor);
    hipLaunchKernelGGL(sweep_sort_scan, dim3(1), block, 0, s, cursor, nb);
    hipLaunchKernelGGL(sweep_sort_scatter, grid, block, 0, s, in_dev, batch, h->c.nsets, cursor, h->order);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    h->order_batch = batch;
    return ISMPC_OK;
}

int ismpc_sweep_info(const ismpc_handle* h, int* n_sets, int* newton_iterations, int* mfma_gemm_launches, double* build_ms)
{
    if (!h) return fail(ISMPC_E_INVALID, "null handle");
    if (n_sets) *n_sets = h->sweep ? h->sw.K : 1;
    if (newton_iterations) *newton_iterations = h->sweep ? h->sw.newton_iters : 0;
    if (mfma_gemm_launches) *mfma_gemm_launches = h->sweep ? h->sw.gemm_launches : 0;
    if (build_ms) *build_ms = h->sweep ? (double)h->sw.build_ms : 0.0;
    return ISMPC_OK;
}

// The device-built tables of one set against the host's long-double build of the same parameters (csrc/ismpc_tables.cpp):
// rel_err[t] = max |device - host| / max |host| for t = 0 H^-1, 1 affine tables (U0,Ua,Ub,SU0,SUa,SUb per pattern), 2 W_p, 3 S W_p,
// 4 Hinv S', 5 S Hinv S', 6 anticipative tails, 7 lane-group layout of the affine tables.
int ismpc_sweep_verify_tables(ismpc_handle* h, int set, double* rel_err)
{
    if (!h || !rel_err) return fail(ISMPC_E_INVALID, "null argument");
    if (!h->sweep || set < 0 || set >= h->sw.K) return fail(ISMPC_E_INVALID, "not a sweep handle, or set out of range");
    ON_DEVICE(h);
    ismpc::Tables t; std::string err;
    int rc = ismpc::build_tables(h->sets[set], h->ftsp.data(), h->t.rows, t, err);
    if (rc != ISMPC_OK) return fail(rc, err);
    const ismpc::SweepSlabs& S = h->sw;
    const int N = t.p.N, NG = S.NG, NTq = ismpc::Tables::NT;
    auto fetch = [&](const double* src, size_t n, std::vector<double>& dst) -> bool { dst.resize(n); return hipMemcpy(dst.data(), src, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess; };
    auto rel = [](const std::vector<double>& d, const std::vector<double>& hst) { double e = 0, m = 0; for (size_t i = 0; i < hst.size(); ++i) { e = std::max(e, std::fabs(d[i] - hst[i])); m = std::max(m, std::fabs(hst[i])); } return m > 0 ? e / m : e; };
    std::vector<double> d, hv;
    if (!fetch(S.X0 + (size_t)set * S.s_mat, S.s_mat, d)) return fail(ISMPC_E_NO_DEVICE, "download failed");
    { std::vector<double> a((size_t)N * N), b((size_t)N * N); for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { a[(size_t)i*N+j] = d[(size_t)i*NG+j]; b[(size_t)i*N+j] = t.Hinv[(size_t)i*t.NP+j]; } rel_err[0] = rel(a, b); }
    if (!fetch(S.vtab + (size_t)set * S.s_vtab, S.s_vtab, d)) return fail(ISMPC_E_NO_DEVICE, "download failed");
    rel_err[1] = rel(d, t.vtab);
    if (!fetch(S.Wt + (size_t)set * S.s_W, S.s_W, d)) return fail(ISMPC_E_NO_DEVICE, "download failed");
    rel_err[2] = rel(d, t.Wt);
    if (!fetch(S.SW + (size_t)set * S.s_W, S.s_W, d)) return fail(ISMPC_E_NO_DEVICE, "download failed");
    rel_err[3] = rel(d, t.SW);
    if (!fetch(S.HSt + (size_t)set * S.s_HS, S.s_HS, d)) return fail(ISMPC_E_NO_DEVICE, "download failed");
    rel_err[4] = rel(d, t.HSt);
    if (!fetch(S.SHSt + (size_t)set * S.s_HS, S.s_HS, d)) return fail(ISMPC_E_NO_DEVICE, "download failed");
    rel_err[5] = rel(d, t.SHSt);
    { std::vector<double> dx, dy; if (!fetch(S.tailx + (size_t)set * S.s_tail, S.s_tail, dx) || !fetch(S.taily + (size_t)set * S.s_tail, S.s_tail, dy)) return fail(ISMPC_E_NO_DEVICE, "download failed");
      rel_err[6] = std::max(rel(dx, t.tailx), rel(dy, t.taily)); }
    rel_err[7] = 0.0;
    for (int pass = 0; pass < (S.vqT2 ? 2 : 1); ++pass) {   // the lane-group layouts (16 lanes, 8 lanes), against the host's re-striding of ITS vtab
        const int lpi = pass == 0 ? 16 : 8, R = quad_R(N, lpi); const size_t npp = (size_t)t.npat + 1;
        hv.assign(npp * R * 3 * lpi * 2, 0.0);
        for (size_t pp = 0; pp < npp; ++pp) for (int r = 0; r < R; ++r) for (int k = 0; k < 3; ++k) for (int li = 0; li < lpi; ++li) {
            const int n = li * R + r; const size_t dst = (((pp * R + r) * 3 + k) * lpi + li) * 2;
            hv[dst] = t.vtab[(pp * 6 + 2 * k) * NTq + n]; hv[dst + 1] = t.vtab[(pp * 6 + 2 * k + 1) * NTq + n];
        }
        if (!fetch(pass == 0 ? S.vqT + (size_t)set * S.s_vqT : S.vqT2 + (size_t)set * S.s_vqT2, pass == 0 ? S.s_vqT : S.s_vqT2, d)) return fail(ISMPC_E_NO_DEVICE, "download failed");
        rel_err[7] = std::max(rel_err[7], rel(d, hv));
    }
    return ISMPC_OK;
}

void ismpc_destroy(ismpc_handle* h)
{
    if (!h) return;
    DeviceGuard guard_(h->device);
    for (void* p : h->dev_allocs) (void)hipFree(p);
    if (h->st_in) (void)hipFree(h->st_in);
    if (h->st_out) (void)hipFree(h->st_out);
    if (h->zseen_host) (void)hipHostFree(h->zseen_host);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->zmark) (void)hipFree(h->zmark);
    if (h->order) (void)hipFree(h->order);
    if (h->zstop) (void)hipFree(h->zstop);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int ismpc_solve_batch_device(ismpc_handle* h, int batch, const ismpc_tick_in* in_dev, ismpc_tick_out* out_dev,
                             double* u_traj, void* stream)
{
    if (!h || batch < 0 || (batch > 0 && (!in_dev || !out_dev))) return fail(ISMPC_E_INVALID, "bad argument");
    ON_DEVICE(h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->timing) HIP_TRY(hipEventRecord(h->ev0, s));
    int rc = launch(h, batch, in_dev, nullptr, out_dev, u_traj, -1, s);
    if (rc != ISMPC_OK) return rc;
    if (h->timing) { HIP_TRY(hipEventRecord(h->ev1, s)); h->timed_pending = true; }
    return ISMPC_OK;
}

int ismpc_solve_batch(ismpc_handle* h, int batch, const ismpc_tick_in* in_host, ismpc_tick_out* out_host)
{
    if (!h || batch < 0 || (batch > 0 && (!in_host || !out_host))) return fail(ISMPC_E_INVALID, "bad argument");
    if (batch == 0) return ISMPC_OK;
    ON_DEVICE(h);
    // A batch of a few instances (the reference's own call: one MPCSolver::solve per control tick) is all latency: the kernel
    // reads its records from, and writes them to, host memory mapped into the device's address space -- two memcpy submissions
    // less than the staged path below.  ISMPC_PINNED=0 switches it off.
    constexpr int PIN_BATCH = 64;
    if (batch <= PIN_BATCH && !h->pin_off) {
        if (!h->pin_in) {
            if (const char* e = std::getenv("ISMPC_PINNED")) h->pin_off = std::atoi(e) == 0;
            if (!h->pin_off) {
                if (hipHostMalloc((void**)&h->pin_in, sizeof(ismpc_tick_in) * PIN_BATCH, hipHostMallocMapped) != hipSuccess ||
                    hipHostMalloc((void**)&h->pin_out, sizeof(ismpc_tick_out) * PIN_BATCH, hipHostMallocMapped) != hipSuccess) {
                    (void)hipGetLastError(); h->pin_off = true;
                    if (h->pin_in) { (void)hipHostFree(h->pin_in); h->pin_in = nullptr; }
                }
            }
        }
        if (!h->pin_off) {
            ismpc_tick_in* din = nullptr; ismpc_tick_out* dout = nullptr;
            HIP_TRY(hipHostGetDevicePointer((void**)&din, h->pin_in, 0));
            HIP_TRY(hipHostGetDevicePointer((void**)&dout, h->pin_out, 0));
            std::memcpy(h->pin_in, in_host, sizeof(ismpc_tick_in) * (size_t)batch);
            const int rc = ismpc_solve_batch_device(h, batch, din, dout, nullptr, h->own_stream);
            if (rc != ISMPC_OK) return rc;
            HIP_TRY(hipStreamSynchronize(h->own_stream));
            std::memcpy(out_host, h->pin_out, sizeof(ismpc_tick_out) * (size_t)batch);
            return ISMPC_OK;
        }
    }
    // zero copy needs device-visible addresses for the caller's records (page-locked AND mapped: hipHostMalloc / hipHostRegister give
    // both under unified addressing); anything else -- also a registration without a device mapping -- takes the staged path
    const ismpc_tick_in* zc_in = nullptr; ismpc_tick_out* zc_out = nullptr;
    bool zero_copy = h->host_mode != 0 && !h->dense_path && host_is_pinned(in_host, sizeof(ismpc_tick_in) * (size_t)batch) && host_is_pinned(out_host, sizeof(ismpc_tick_out) * (size_t)batch);
    if (zero_copy && (hipHostGetDevicePointer((void**)&zc_in, const_cast<ismpc_tick_in*>(in_host), 0) != hipSuccess ||
                      hipHostGetDevicePointer((void**)&zc_out, out_host, 0) != hipSuccess)) { (void)hipGetLastError(); zero_copy = false; }
    if (batch > h->st_cap && !(zero_copy && h->host_mode == 3)) {      // device staging (not needed when both sides are in place)
        if (h->st_in) (void)hipFree(h->st_in);
        if (h->st_out) (void)hipFree(h->st_out);
        h->st_in = nullptr; h->st_out = nullptr; h->st_cap = 0;
        HIP_TRY(hipMalloc((void**)&h->st_in, sizeof(ismpc_tick_in) * (size_t)batch));
        HIP_TRY(hipMalloc((void**)&h->st_out, sizeof(ismpc_tick_out) * (size_t)batch));
        h->st_cap = batch;
    }
    // Page-locked caller buffers (hipHostMalloc / hipHostRegister / ismpc_host_alloc / ismpc_host_register): ZERO COPY -- the kernel
    // reads the records from, and writes them to, the caller's memory over PCIe.  Reads and writes travel in opposite directions
    // at the same time and no DMA submission sits in front of or behind the launch.  Measured on MI355X at 65 536 records (4.7 MB
    // in, 5.2 MB out; scripts/host_path_probe.py): 0.193 ms per call against 0.257 ms for DMA in -> kernel -> DMA out from the same
    // buffers; chunking that pipeline over two streams gains nothing (0.243 ms with 2 chunks, slower with more: each DMA
    // submission costs ~10 us), nor do non-coherent / write-combined allocations or 64-byte-line stores.  ISMPC_HOST_MODE: bit 0 =
    // read in place, bit 1 = write in place (default 3; 0 = the staged path).  Same kernel, same records, bit for bit.
    // Pageable buffers take the staged path below (the HIP runtime stages them).
    if (zero_copy) {
        const ismpc_tick_in* din = h->st_in; ismpc_tick_out* dout = h->st_out;
        hipStream_t s = h->own_stream;
        if (h->host_mode & 1) din = zc_in;
        else HIP_TRY(hipMemcpyAsync(h->st_in, in_host, sizeof(ismpc_tick_in) * (size_t)batch, hipMemcpyHostToDevice, s));
        if (h->host_mode & 2) dout = zc_out;
        const int rc = ismpc_solve_batch_device(h, batch, din, dout, nullptr, s);
        if (rc != ISMPC_OK) return rc;
        if (!(h->host_mode & 2)) HIP_TRY(hipMemcpyAsync(out_host, h->st_out, sizeof(ismpc_tick_out) * (size_t)batch, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return ISMPC_OK;
    }
    hipStream_t s = h->own_stream;
    HIP_TRY(hipMemcpyAsync(h->st_in, in_host, sizeof(ismpc_tick_in) * (size_t)batch, hipMemcpyHostToDevice, s));
    int rc = ismpc_solve_batch_device(h, batch, h->st_in, h->st_out, nullptr, s);
    if (rc != ISMPC_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, h->st_out, sizeof(ismpc_tick_out) * (size_t)batch, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ISMPC_OK;
}

int ismpc_rollout_device(ismpc_handle* h, int batch, ismpc_tick_in* state_dev, int first_frame, int ticks,
                         ismpc_tick_out* traj_dev, void* stream)
{
    if (!h || batch < 0 || ticks < 0 || first_frame < 0 || (batch > 0 && !state_dev)) return fail(ISMPC_E_INVALID, "bad argument");
    ON_DEVICE(h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    StreamMark mark_{h, s};
    if (h->timing) HIP_TRY(hipEventRecord(h->ev0, s));
    if (batch > 0 && ticks > 0 && h->kernel_rollout && !h->dense_path && h->quad_path && h->c.N <= 128 && h->z_fallback) {
        // the whole closed loop in ONE launch: state in registers, one trajectory record per tick (ismpc_rollout_quad)
        const LaneLayout lay = pick_layout(h, batch, false);
        const int lpi = lay.lpi, RQ = quad_R(h->c.N, lpi);
        DevConst cq = h->c;
        cq.vqT = lay.vqT; cq.tzgT = lay.tzgT;
        const int waves = (batch * lpi + 63) / 64;
        const dim3 qgrid((waves + ISMPC_QUAD_WAVES - 1) / ISMPC_QUAD_WAVES), qblock(64 * ISMPC_QUAD_WAVES);
        if (batch > h->zstop_cap) {                       // stream-ordered growth, as zmark (ismpc_reserve sizes it beforehand)
            HIP_TRY(grow_sync(h, s));
            if (h->zstop) HIP_TRY(hipFreeAsync(h->zstop, s));
            h->zstop = nullptr; h->zstop_cap = 0;
            HIP_TRY(hipMallocAsync((void**)&h->zstop, sizeof(int) * (size_t)batch, s));
            h->zstop_cap = batch;
        }
        const int lid = ++h->launch_id;
        const dim3 rgrid(std::min((batch + ISMPC_QUAD_WAVES - 1) / ISMPC_QUAD_WAVES, 64));
#define ISMPC_ROLL(RR, LL, RW_) do { \
        hipLaunchKernelGGL((ismpc_rollout_quad<RR, LL, RW_, false>), qgrid, qblock, 0, s, cq, state_dev, traj_dev, batch, first_frame, ticks, h->zstop, lid); \
        hipLaunchKernelGGL((ismpc_rollout_quad<RR, LL, RW_, true>), rgrid, qblock, 0, s, cq, state_dev, traj_dev, batch, first_frame, ticks, h->zstop, lid); } while (0)
        if (h->sweep) {
#define ISMPC_ROLLS(RR, RW_) do { \
        hipLaunchKernelGGL((ismpc_rollout_quad<RR, 16, RW_, false, true>), qgrid, qblock, 0, s, cq, state_dev, traj_dev, batch, first_frame, ticks, h->zstop, lid); \
        hipLaunchKernelGGL((ismpc_rollout_quad<RR, 16, RW_, true, true>), rgrid, qblock, 0, s, cq, state_dev, traj_dev, batch, first_frame, ticks, h->zstop, lid); } while (0)
            if (RQ == 4) ISMPC_ROLLS(4, 1); else if (RQ == 7) ISMPC_ROLLS(7, 2); else ISMPC_ROLLS(8, 2);
#undef ISMPC_ROLLS
        } else {
        ISMPC_SHAPES(ISMPC_ROLL)
        }
#undef ISMPC_ROLL
        HIP_TRY(hipGetLastError());
    } else {
        for (int t = 0; t < ticks; ++t) {
            int rc = launch(h, batch, nullptr, state_dev, traj_dev ? traj_dev + (size_t)t * batch : nullptr, nullptr, first_frame + t, s);
            if (rc != ISMPC_OK) return rc;
        }
    }
    if (h->timing) { HIP_TRY(hipEventRecord(h->ev1, s)); h->timed_pending = true; }
    return ISMPC_OK;
}

// Page-locked host memory for callers without HIP headers (the pipelined ismpc_solve_batch needs it on both sides).
int ismpc_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return fail(ISMPC_E_INVALID, "bad argument");
    *out = nullptr;
    unsigned flags = hipHostMallocDefault;
    if (const char* e = std::getenv("ISMPC_HOST_ALLOC_FLAGS")) flags = (unsigned)std::strtoul(e, nullptr, 0);
    if (hipHostMalloc(out, bytes, flags) != hipSuccess) { (void)hipGetLastError(); return fail(ISMPC_E_ALLOC, "hipHostMalloc failed"); }
    return ISMPC_OK;
}
int ismpc_host_free(void* p)
{
    if (!p) return ISMPC_OK;
    HIP_TRY(hipHostFree(p));
    return ISMPC_OK;
}
int ismpc_host_register(void* p, size_t bytes)
{
    if (!p || bytes == 0) return fail(ISMPC_E_INVALID, "bad argument");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return ISMPC_OK;
}
int ismpc_host_unregister(void* p)
{
    if (!p) return fail(ISMPC_E_INVALID, "bad argument");
    HIP_TRY(hipHostUnregister(p));
    return ISMPC_OK;
}

int ismpc_reserve(ismpc_handle* h, int max_batch)
{
    if (!h || max_batch < 0) return fail(ISMPC_E_INVALID, "bad argument");
    ON_DEVICE(h);
    if (h->z_fallback && max_batch > h->zmark_cap) {
        if (h->zmark) HIP_TRY(hipFree(h->zmark));
        h->zmark = nullptr; h->zmark_cap = 0;
        HIP_TRY(hipMalloc((void**)&h->zmark, zscratch_bytes(max_batch)));
        h->zmark_cap = max_batch;
    }
    if (max_batch > h->zstop_cap) {
        if (h->zstop) HIP_TRY(hipFree(h->zstop));
        h->zstop = nullptr; h->zstop_cap = 0;
        HIP_TRY(hipMalloc((void**)&h->zstop, sizeof(int) * (size_t)max_batch));
        h->zstop_cap = max_batch;
    }
    return ISMPC_OK;
}

#ifdef ISMPC_STAMPS
int ismpc_debug_stamps(unsigned long long* dst, int reset)
{
    if (dst && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16384 * 8) != hipSuccess) return -2;
    if (reset) { std::vector<unsigned long long> z(16384 * 8, 0ull); if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z.data(), z.size() * 8) != hipSuccess) return -2; }
    return 0;
}
#endif

int ismpc_fallback_counters(ismpc_handle* h, int* out4)
{
    if (!h || !out4) return fail(ISMPC_E_INVALID, "null argument");
    ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out4, h->c.zflag, 4 * sizeof(int), hipMemcpyDeviceToHost));
    return ISMPC_OK;
}

int ismpc_get_params(const ismpc_handle* h, ismpc_params* out)
{
    if (!h || !out) return fail(ISMPC_E_INVALID, "null argument");
    *out = h->t.p; return ISMPC_OK;
}
int ismpc_midpoint_rows(const ismpc_handle* h) { return h ? h->t.nmid : ISMPC_E_INVALID; }
int ismpc_get_midpoint(const ismpc_handle* h, double* dst, int capacity_rows)
{
    if (!h || !dst || capacity_rows < h->t.nmid) return fail(ISMPC_E_INVALID, "bad argument");
    for (int i = 0; i < h->t.nmid; ++i) { dst[3*i] = h->t.midx[i]; dst[3*i+1] = h->t.midy[i]; dst[3*i+2] = h->t.midz[i]; }
    return ISMPC_OK;
}
int ismpc_set_timing(ismpc_handle* h, int enabled)
{
    if (!h) return fail(ISMPC_E_INVALID, "null handle");
    h->timing = enabled != 0; h->timed_pending = false; return ISMPC_OK;
}
double ismpc_last_kernel_ms(ismpc_handle* h)
{
    if (!h || !h->timing) return 0.0;
    if (h->timed_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(h->ev1) == hipSuccess && hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_ms = ms;
        h->timed_pending = false;
    }
    return h->last_ms;
}

}  // extern "C"
