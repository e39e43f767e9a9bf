// Wavefront-per-QP kernel of Formulation A: the STRUCTURED dual active-set solver (docs/models/proto_structured.py is its numpy
// model).  Included by one translation unit per rows-per-lane value (ismpc_a_wave_rl{2,3,4}.hip).
//
// The per-axis QP of one instance (walking/quad_walk_no_plots.m:153-293):
//     min 1/2 |u|^2 + Qf/2 |f - p|^2      u = ZMP velocities (C), f = footsteps (F)
//     s.t. a'u = b                          stability (anticipative tail)          (:227-242)
//          lo_i <= dt cumsum(u)_i - M_i f <= hi_i     ZMP band around the mapped footstep  (:153-181)
//          -bl_r <= f_r - f_{r-1} <= bu_r             kinematic                           (:187-222)
//
// No matrix over the working set exists.  For the active ZMP rows, sorted by sample index i_1 < i_2 < ..., the Gram block in
// the H^-1 metric is dt^2 min(i_j, i_k) + (footstep coupling)/Qf.  dt^2 min(.,.) is the covariance of a random walk, so its
// inverse is TRIDIAGONAL: (K^-1 y)_j = (y_j - y_prev)/g_j - (y_next - y_j)/g_next with g the index gaps -- each active row
// only needs its previous / next active row -- and K^-1 applied to a kernel column min(i+, .) is linear interpolation at i+
// (two non-zeros).  Everything else -- the footstep coupling M~ (F columns), the stability row and the F kinematic rows -- is
// a rank <= 2F+1 border: V_i = [M~_i, dt PA_i, Bk_i] has a closed form per row, G = V' K^-1 V / dt^2 (m x m, m = 2F+1) is kept
// by +/- one outer product per gap created/destroyed, and one quasi-definite m x m system per step gives the border unknowns.
// Step lengths, add / drop logic and termination are Goldfarb-Idnani's.  Lanes own RL consecutive rows (row = ZMP sample), so
// the primal direction is "own multiplier as an impulse + one suffix scan" with no scatter; wave collectives are DPP scans.
//
// Coordinates: every position of the QP (ZMP, footsteps, plan) is taken RELATIVE TO THE CURRENT FOOTSTEP.  The mapping rows sum
// to one, so the band of every row becomes the same pair of numbers (-(zmp - cur) -+ w/2), the first kinematic row becomes
// symmetric, and all quantities the solver touches are step-sized: that is what lets the same code run in fp32 (Real = float:
// right-hand sides are formed in fp64 and rounded once; the LIP update of the state stays fp64) as well as in fp64.
//
// Route of one solve (docs/models/proto_passes.py): two exact Goldfarb-Idnani steps from the equality-only point, block
// passes (every violated row enters at once; a negative run end leaves with the rows whose lumped multipliers stay <= 0; one
// structured solve per pass), rounds of (exact steps, passes) while rows stay violated, Goldfarb-Idnani to the end, and a check
// of the returned point.  Closed loops start the passes from the previous tick's working set.  An fp32 solve whose block solve
// fails its check is handed to the fp64 instantiation (defer_list).
//
// Register budget: per row a lane keeps u, the mapping weight, the multiplier (Real), a float norm, and two packed ints
// (first mapped footstep + state; previous / next active row).  The bounds are two wave-uniform numbers.
#pragma once
#include <cmath>
#include <algorithm>
#include <type_traits>
#include "ismpc_a_dev.hpp"

namespace ismpc_a {

// -DISMPC_A_PHASES (diagnostic build, scripts/phases_a.py): shader-clock time of every wavefront, split by solver phase and summed
// over the launch (s_memtime deltas, wave-uniform accumulators, one atomicAdd per phase per QP).  Nothing of it exists in the product build.
#ifdef ISMPC_A_PHASES
constexpr int NPH = 16;
static __device__ unsigned long long g_phase[NPH];
#define PH_DECL unsigned long long ph_t_ = __builtin_amdgcn_s_memtime(); unsigned ph_acc_[NPH]; for (int k_ = 0; k_ < NPH; ++k_) ph_acc_[k_] = 0u
// the time since the previous mark belongs to phase k_ (a compile-time constant: the accumulators stay in scalar registers)
#define PH(k_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc_[k_] += (unsigned)(t_ - ph_t_); ph_t_ = t_; } while (0)
#define PH_FLUSH() do { if (lane == 0) { _Pragma("unroll") for (int k_ = 0; k_ < NPH; ++k_) if (ph_acc_[k_]) atomicAdd(&g_phase[k_], (unsigned long long)ph_acc_[k_]); } _Pragma("unroll") for (int k_ = 0; k_ < NPH; ++k_) ph_acc_[k_] = 0u; } while (0)
#else
#define PH_DECL do {} while (0)
#define PH(k_) do {} while (0)
#define PH_FLUSH() do {} while (0)
#endif

// ---- precision traits ---------------------------------------------------------------------------------------------------
template <typename R> struct Num;
template <> struct Num<double> {
    static constexpr double viol_rel = 1e-11, viol_abs = 1e-13;     // a row is violated beyond  rel (|v| + |bounds|) + abs
    static constexpr double bound_rel = 1e-6, bound_abs = 1e-8;     // block solve: active rows must sit on their bounds (a breakdown
    static constexpr double eq_rel = 1e-6;                          // is off by orders of magnitude more) and the stability row must hold
    static constexpr double gamma_rel = 1e-12;                      // full step possible when gamma > rel |n+|^2
    static constexpr double final_rel = 1e-6, final_abs = 1e-7;     // check of the returned point (band half-width: 1e-2)
    static constexpr double mult_rel = 1e-8;                        // polish: multipliers may be this negative
};
template <> struct Num<float> {
    static constexpr float viol_rel = 2e-6f, viol_abs = 2e-6f;
    static constexpr float bound_rel = 1e-4f, bound_abs = 1e-5f;
    static constexpr float eq_rel = 1e-4f;
    static constexpr float gamma_rel = 2e-6f;
    static constexpr float final_rel = 2e-4f, final_abs = 2e-5f;
    static constexpr float mult_rel = 1e-4f;
};

// ---- wave primitives, both precisions -------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ double dppv(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ float dppv(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, BOUND_ZERO));
}
template <int CTRL, int RM> __device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xf, false); }

template <typename R> __device__ __forceinline__ R wave_scan_up(R v)       // inclusive prefix sum over the 64 lanes
{
    v += dppv<0x111, 0xf, true>(R(0), v);
    v += dppv<0x112, 0xf, true>(R(0), v);
    v += dppv<0x114, 0xf, true>(R(0), v);
    v += dppv<0x118, 0xf, true>(R(0), v);
    v += dppv<0x142, 0xa, false>(R(0), v);
    v += dppv<0x143, 0xc, false>(R(0), v);
    return v;
}
__device__ __forceinline__ double rl(double v, int l)
{
    const int ll = __builtin_amdgcn_readfirstlane(l);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), ll), __builtin_amdgcn_readlane(__double2loint(v), ll));
}
__device__ __forceinline__ float rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(l))); }
__device__ __forceinline__ int rl(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
template <typename R> __device__ __forceinline__ R wave_sum(R v) { return rl(wave_scan_up(v), 63); }
template <typename R> __device__ __forceinline__ R wave_min(R v)
{
    const R inf = R(INFINITY);
    v = fmin(v, dppv<0x111, 0xf, false>(inf, v)); v = fmin(v, dppv<0x112, 0xf, false>(inf, v));
    v = fmin(v, dppv<0x114, 0xf, false>(inf, v)); v = fmin(v, dppv<0x118, 0xf, false>(inf, v));
    v = fmin(v, dppv<0x142, 0xa, false>(inf, v)); v = fmin(v, dppv<0x143, 0xc, false>(inf, v));
    return rl(v, 63);
}
__device__ __forceinline__ int wave_scan_max_i(int v)      // inclusive prefix maximum over the 64 lanes
{
    const int lo = -2147483647 - 1;
    v = max(v, dpp_i<0x111, 0xf>(lo, v)); v = max(v, dpp_i<0x112, 0xf>(lo, v)); v = max(v, dpp_i<0x114, 0xf>(lo, v));
    v = max(v, dpp_i<0x118, 0xf>(lo, v)); v = max(v, dpp_i<0x142, 0xa>(lo, v)); v = max(v, dpp_i<0x143, 0xc>(lo, v));
    return v;
}
// 1/x to rounding error: hardware reciprocal + Newton (the IEEE division sequence is more than twice as long, and the
// active-set loop divides by gaps and pivots on its critical path)
__device__ __forceinline__ double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float frcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    return r;
}
__device__ __forceinline__ float rsq_f(float x) { return __builtin_amdgcn_rsqf(x); }

// Sums of K per-lane values over the 64 lanes (wave_fold_sums below).  Values are folded pairwise: after a step with partner
// lane ^ 2^s a lane keeps half of its values (the half its partner sent sums for), so K values cost about K + 6 exchange-adds
// instead of 6 K.
template <typename R, int XOR> __device__ __forceinline__ R lane_xor(R v);
template <> __device__ __forceinline__ float lane_xor<float, 1>(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, false)); }
template <> __device__ __forceinline__ float lane_xor<float, 2>(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, false)); }
template <> __device__ __forceinline__ float lane_xor<float, 4>(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F)); }   // xor_mask 4, and_mask 0x1f
template <> __device__ __forceinline__ float lane_xor<float, 8>(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xf, 0xf, false)); }   // row_ror:8
template <> __device__ __forceinline__ float lane_xor<float, 16>(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }  // xor_mask 16
template <> __device__ __forceinline__ float lane_xor<float, 32>(float v) { return __shfl_xor(v, 32); }
template <int XOR> __device__ __forceinline__ double lane_xor_d(double v)
{
    return __hiloint2double(__float_as_int(lane_xor<float, XOR>(__int_as_float(__double2hiint(v)))),
                            __float_as_int(lane_xor<float, XOR>(__int_as_float(__double2loint(v)))));
}
template <> __device__ __forceinline__ double lane_xor<double, 1>(double v) { return lane_xor_d<1>(v); }
template <> __device__ __forceinline__ double lane_xor<double, 2>(double v) { return lane_xor_d<2>(v); }
template <> __device__ __forceinline__ double lane_xor<double, 4>(double v) { return lane_xor_d<4>(v); }
template <> __device__ __forceinline__ double lane_xor<double, 8>(double v) { return lane_xor_d<8>(v); }
template <> __device__ __forceinline__ double lane_xor<double, 16>(double v) { return lane_xor_d<16>(v); }
template <> __device__ __forceinline__ double lane_xor<double, 32>(double v) { return lane_xor_d<32>(v); }

// N values per lane (N a power of two), partner lane ^ X: a lane whose X bit is clear keeps the even members, the other the
// odd ones; each kept member becomes own + partner's.  After log2 N steps one value is left: the sum, over the 2^log2 N lanes
// that differ in the bits used so far, of member (lane & (N0 - 1)) of the original N0 values.
template <typename R, int N, int X>
__device__ __forceinline__ R fold_pow2(const R* v, const int lane)
{
    if constexpr (N == 1) return v[0];
    else {
        const bool hi = (lane & X) != 0;
        R w[N / 2];
#pragma unroll
        for (int j = 0; j < N / 2; ++j) {
            const R keep = hi ? v[2 * j + 1] : v[2 * j], send = hi ? v[2 * j] : v[2 * j + 1];
            w[j] = keep + lane_xor<R, X>(send);
        }
        return fold_pow2<R, N / 2, X * 2>(w, lane);
    }
}
template <typename R, int X> __device__ __forceinline__ R finish_sum(R v)     // plain butterfly over the remaining lane bits
{
    if constexpr (X <= 32) { v += lane_xor<R, X>(v); return finish_sum<R, X * 2>(v); }
    else return v;
}
// out[t] = sum over the wavefront of acc[t], t < NS (NS <= 63): the binary digits of NS give chunks of 32 / 16 / ... / 1 values
template <typename R, int NS, int P, int OFF>
__device__ __forceinline__ void wave_fold_sums_from(const R* acc, R* out, const int lane)
{
    if constexpr (P >= 0) {
        constexpr int N = 1 << P;
        if constexpr ((NS & N) != 0) {
            const R part = fold_pow2<R, N, 1>(acc + OFF, lane);
            const R tot = finish_sum<R, N>(part);
            if (lane < N) out[OFF + lane] = tot;
            wave_fold_sums_from<R, NS, P - 1, OFF + N>(acc, out, lane);
        } else wave_fold_sums_from<R, NS, P - 1, OFF>(acc, out, lane);
    }
}
template <typename R, int NS> __device__ __forceinline__ void wave_fold_sums(const R* acc, R* out, const int lane)
{
    static_assert(NS >= 1 && NS < 64, "chunk decomposition covers 1..63 sums");
    wave_fold_sums_from<R, NS, 5, 0>(acc, out, lane);
}

// a flag carried in the last mantissa bit of a value (one ulp of the value is given up)
__device__ __forceinline__ float with_flag(float v, bool f) { return __int_as_float((__float_as_int(v) & ~1) | (f ? 1 : 0)); }
__device__ __forceinline__ double with_flag(double v, bool f) { return __longlong_as_double((__double_as_longlong(v) & ~1ll) | (f ? 1ll : 0ll)); }
__device__ __forceinline__ bool flag_of(float v) { return (__float_as_int(v) & 1) != 0; }
__device__ __forceinline__ bool flag_of(double v) { return (__double_as_longlong(v) & 1ll) != 0; }

#ifndef ISMPC_A_GCAP
#define ISMPC_A_GCAP 64
#endif
constexpr int GCAP = ISMPC_A_GCAP;     // Gram sums of a block solve accumulated per pass (a power of two; 64 = all at once, as rounds 1-3 did)

// ---- per-wavefront LDS.  Small vectors (length m = 2F+1 or F+2) are kept ONE ELEMENT PER LANE in registers and mirrored here
// when other lanes need them by index; nothing of size "working set" is stored anywhere.
template <typename R, int F> struct WaveLds {
    static constexpr int m = 2 * F + 1;
    static constexpr int NTH = F * (F + 1) / 2 + 2 * F + 2;
    R     sv[WG];                      // V_row . y of every ZMP row
    R     w1s[WG];                     // mapping weight of every row (wave-uniform look-ups by row index)
    unsigned char k1s[WG];             // first mapped footstep of every row
    R     comb[F + 2];                 // footstep-column coefficients seen by a row: comb[k1], comb[k1+1]
    R     fl[F + 2];                   // f[0..F+1] with fl[0] = fl[F+1] = 0 (relative to the current footstep)
    R     pf[F + 2];                   // the plan's footsteps (same layout): block warm start
    R     th[NTH];                     // Gram sums of the block warm start: Theta (upper triangle), psi, gamma, sigma, gamma_E
    R     G[m * m];                    // V' K^-1 V / dt^2 over the active ZMP rows
    R     vp[m], hx[m], d1[m], d2[m], d0[m], cc[m], mt[m];
};
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// value held by the owner of ZMP row `row` (1-based, wave-uniform) in a per-row register array.  Every element is read from the owner
// lane and the wave-uniform position picks among the SCALARS: written as a select among the lane's own registers followed by one
// readlane, the compiler turned the (uniform) index into an index into a private array -- three scratch stores and a dependent scratch
// load per call, which was all of the scratch traffic of the shapes that spill nothing (7-15x the algorithmic HBM bytes, round 3).
template <typename V, int RL> __device__ __forceinline__ V at_row(const V (&v)[RL], int row)
{
    const int o = (row - 1) / RL, k = (row - 1) - o * RL;
    V x = rl(v[0], o);
#pragma unroll
    for (int r = 1; r < RL; ++r) { const V y = rl(v[r], o); x = (k == r) ? y : x; }
    return x;
}
// b^n, 0 <= n < 2048, by repeated squaring (per-instance gait parameters: every weight of the stability row and of the
// anticipative tail is a power of lambda = exp(-eta dt); one exp per QP instead of seven, a pow, a cosh and a sinh)
__device__ __forceinline__ double ipow(double b, int n)
{
    double r = 1.0;
#pragma unroll
    for (int k = 0; k < 11; ++k) { if (n & 1) r *= b; b *= b; n >>= 1; }
    return r;
}

// lam^n for a per-lane n < 64 (six bits); *b64 = lam^64, what the squarings end on
__device__ __forceinline__ double ipow6(double b, int n, double* b64)
{
    double r = 1.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { if (n & 1) r *= b; b *= b; n >>= 1; }
    if (b64) *b64 = b;
    return r;
}
// a wave-uniform value pinned to scalar registers (kept across QPs: a spilled scalar register costs a v_readlane, not scratch)
__device__ __forceinline__ double uni(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// element `e` (this lane's) of the border row V = [M~ (F), dt PA, Bk (F)] of a ZMP row with mapping (k1, w1, 1-w1), PA = pa
template <typename R, int F> __device__ __forceinline__ R border_elem(int e, int k1, R w1, R pa, R dt, R isq)
{
    if (e == F) return dt * pa;
    const R w2 = R(1) - w1;
    const int r = e < F ? e + 1 : e - F;                  // footstep column 1..F
    const R mr = (r == k1) ? w1 : ((r == k1 + 1) ? w2 : R(0));
    if (e < F) return mr * isq;
    const R mp = (r - 1 >= 1) ? ((r - 1 == k1) ? w1 : ((r - 1 == k1 + 1) ? w2 : R(0))) : R(0);
    return (-mr + mp) * isq;
}

// centreline value cl(k0+1) without the table: quad_walk_no_plots.m:86-99 (initial structure) / :540-549 (rebuilt one),
// linspace as MATLAB evaluates it.  Used when step / ds differ per instance.
__device__ __forceinline__ double cl_closed(const double* __restrict__ fs, int step, float rstep, int ds, double inv_dsm1, bool rebuilt, int k0)
{
    int s = (int)((float)k0 * rstep);                       // k0 / step without the integer-division sequence (rstep = 1 / step)
    if (s * step > k0) --s; else if ((s + 1) * step <= k0) ++s;
    const int r = k0 - s * step, q = r - (step - ds);
    const double d1 = fs[s];
    if (q <= 0 || (rebuilt && s == 0)) return d1;
    const double d2 = fs[s + 1];
    if (q == ds - 1) return d2;
    return d1 + ((double)q * (d2 - d1)) * inv_dsm1;         // inv_dsm1 = 1 / (ds - 1)
}

// Residency target (wavefronts per SIMD = workgroups per CU, 256 threads each).  Measured with scripts/occ_sweep.sh on the
// round-2 solver (MI355X, 16 384 pushed instances) after the lane masks stopped living in spilled scalar registers: fp32 4 for
// three rows per lane (128 registers, 5 spilled: walk C=150 4.7e7 against 4.1e7 ticks/s at 3, trot C=160 4.2e7 against 3.7e7), 3
// and for four (Monte-Carlo: 2.65e7 against 2.54e7 at 3, 1.95e7 at 2; 55 spilled), 3 for two rows per lane (4 changes nothing; four rows per lane -- the Monte-Carlo shape -- 2.54e7 ticks/s at 3
// against 1.95e7 at 2: 14 spilled registers now, 160 before); fp64 3 for two rows per lane (walk C=100: 5.2e7 against 4.3e7 at 2,
// 8 spilled registers) and for three (walk C=150: 3.44e7 against 3.0e7 at 2, trot C=160 3.0e7 against 2.55e7, with 50 spilled
// registers); 2 for four rows per lane and for per-instance parameters (LDS).
#ifndef ISMPC_A_PAIR_SETUP       // 0: every QP builds its instance's set-up itself (A/B knob)
#define ISMPC_A_PAIR_SETUP 1
#endif
#ifndef ISMPC_A_OCC_F32_RL2      // tuning knobs (scripts/occ_sweep.sh builds variants)
#define ISMPC_A_OCC_F32_RL2 3
#endif
#ifndef ISMPC_A_OCC_F32_RL3
#define ISMPC_A_OCC_F32_RL3 4
#endif
#ifndef ISMPC_A_OCC_F32_RL4
#define ISMPC_A_OCC_F32_RL4 4
#endif
#ifndef ISMPC_A_OCC_F64_RL2
#define ISMPC_A_OCC_F64_RL2 3
#endif
#ifndef ISMPC_A_OCC_F64_RL3
#define ISMPC_A_OCC_F64_RL3 3
#endif
#ifndef ISMPC_A_OCC_F64_RL4
#define ISMPC_A_OCC_F64_RL4 2
#endif
template <typename R, int RL, int F, bool PI> constexpr int wave_min_blocks()
{
    if (sizeof(R) == 8 && RL <= 2 && (PI || F > 4)) return 2;   // per-instance: 56-62 KB of LDS per workgroup, two fit a CU whatever the
                                                                // registers allow; five and six footsteps: 29 / 61 spilled registers at 3
    return sizeof(R) == 4 ? (RL <= 2 ? ISMPC_A_OCC_F32_RL2 : (RL == 3 ? ISMPC_A_OCC_F32_RL3 : ISMPC_A_OCC_F32_RL4))
                          : (RL <= 2 ? ISMPC_A_OCC_F64_RL2 : ((RL == 3 && !PI) ? ISMPC_A_OCC_F64_RL3 : ISMPC_A_OCC_F64_RL4));
}

// RL = ZMP rows per lane (C <= 64 RL), F = footsteps in the horizon (m = 2F+1 border columns).
// PI: per-instance gait parameters (ismpc_a_inst): height, Qf, step, ds, F <= the template F, base plan.
template <typename R, int RL, int F, bool PI>
__global__ __launch_bounds__(WG, (wave_min_blocks<R, RL, F, PI>()))
void ismpc_a_tick_wave(const DevA* __restrict__ cp, const ismpc_a_state* __restrict__ state_in, ismpc_a_state* __restrict__ state,
                       const ismpc_a_inst* __restrict__ ipar, const double* __restrict__ push, ismpc_a_out* __restrict__ out, int batch,
                       int* __restrict__ work_counter, unsigned long long* __restrict__ hist, int hist_load,
                       const int* __restrict__ order, const int* __restrict__ count_ptr, const int claim_chunk_, const int static_q,
                       const int order_is_qp, int* __restrict__ defer_list, int* __restrict__ defer_count, const PiPre* __restrict__ pre)
{
    // the re-solve launch behind an fp32 launch that handed nothing over (the usual case) leaves before it builds its tables
    if (order_is_qp && *count_ptr == 0) return;
    using NM = Num<R>;
    const DevA& c = *cp;                                   // handle constants, read from memory where they are used (by value they would
    constexpr int m = 2 * F + 1;                           // sit in 70 scalar registers for the whole persistent loop)
    __shared__ WaveLds<R, F> lds_all[WG / 64];
    __shared__ R a_s[PI ? 1 : WG], pa_s[PI ? 1 : WG + 1];  // stability row and its prefix sums: same for every QP of the handle
    __shared__ R a_pi[PI ? WG / 64 : 1][PI ? WG : 1], pa_pi[PI ? WG / 64 : 1][PI ? WG + 1 : 1];   // ... or one per wavefront
    // the same prefix sums and those of a^2 in fp64, whatever the precision of the solve: the two places where the stability row
    // is (nearly) in the span of the active ZMP rows are evaluated from them without cancellation (see gap_terms below)
    __shared__ R rinv[WG + 1];                              // 1 / g for the index gaps g = 1 .. C between active rows (a read instead of a division)
    // G = V'K^-1 V / dt^2 from the Gram sums of a block solve: entry e = (i, j) of the m x m matrix is scale(e) x (at most four signed
    // entries of L.th).  Which ones is a property of (e, F) alone: tabulated once per workgroup -- evaluated in place it was ~110 VALU
    // instructions and four dependent LDS round trips per entry, on every block solve.
    __shared__ unsigned int gdesc[m * m];                   // four bytes per entry: L.th index (bits 0-5) | sign as a 2-bit integer (6-7; 0: unused)
    __shared__ unsigned char gkind[m * m];                  // bits 0-1 scale: 0 = 1, 1 = 1 / sqrt(Qf), 2 = 1 / Qf; bits 2-4: the CONSTANT part of the
                                                            // small system at this entry as a 3-bit integer (+1 on the footstep diagonal, -S_xx on the
                                                            // kinematic block: -2 / -1 on its diagonal, +1 beside it) -- L.G holds G plus that constant,
                                                            // so the solve reads its matrix instead of re-deriving the pattern per entry  (5 bytes per
                                                            // entry in all: the per-instance fp32 shape sits 80 bytes under the LDS of four workgroups per CU)
    __shared__ double pad_s[PI ? 1 : WG + 1], pa2d_s[PI ? 1 : WG + 1];
    __shared__ double pad_pi[PI ? WG / 64 : 1][PI ? WG + 1 : 1], pa2d_pi[PI ? WG / 64 : 1][PI ? WG + 1 : 1];
    // `lane` is re-declared opaque (LANE_FRESH) at the head of every solver phase: comparisons against it (lane == k, lane < m, the
    // fold masks ...) are loop invariants, the compiler hoists dozens of 64-bit masks out of the persistent loop, they do not fit
    // the scalar register file and every use then restores its mask from a spill VGPR lane by lane (two v_readlane and a wait
    // state against one v_cmp to recompute it)
    int lane = threadIdx.x & 63;
#define LANE_FRESH() asm volatile("" : "+v"(lane))
#define G_CONST(kd_) ((R)(((int)((unsigned)(kd_) << 27)) >> 29))      /* bits 2-4 of gkind, sign-extended */
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds<R, F>& L = lds_all[wv];
    const int C = c.C, P = c.P;
    const R dt = (R)c.dt, idt = (R)(1.0 / c.dt), idt2 = (R)(1.0 / (c.dt * c.dt));
    const bool klane = lane >= 1 && lane <= F;            // lane r owns kinematic row r (and f_r)
    for (int k = threadIdx.x; k <= WG; k += WG) rinv[k] = k > 0 ? (R)(1.0 / (double)k) : R(0);
    if (threadIdx.x == 0) rinv[WG] = (R)(1.0 / (double)WG);
    for (int e = threadIdx.x; e < m * m; e += WG) {
        constexpr int NT_ = F * (F + 1) / 2;
        // Phi(e): e < F -> +col e+1 ; e > F -> -col (e-F) + col (e-F-1) [if >= 1]; Theta(r, q) (1-based, symmetric) sits at TH(r, q) of L.th
        auto TH = [](int r, int q) { const int lo_ = r < q ? r : q, hi_ = r < q ? q : r; return (lo_ - 1) * F - ((lo_ - 1) * (lo_ - 2)) / 2 + (hi_ - lo_); };
        const int i_ = e / m, j_ = e - i_ * m;
        const int ra = i_ < F ? i_ + 1 : i_ - F, rb = j_ < F ? j_ + 1 : j_ - F;     // leading column of Phi(e)
        const int sa = i_ < F ? 1 : -1, sb = j_ < F ? 1 : -1;
        const bool a2 = i_ > F && ra >= 2, b2 = j_ > F && rb >= 2;                   // second term: +col (r-1)
        int id0 = 0, id1 = 0, id2 = 0, id3 = 0, cf0 = 0, cf1 = 0, cf2 = 0, cf3 = 0, kd;      // scalars, not arrays: a dynamically indexed
        if (i_ == F && j_ == F) { id0 = NT_ + 2 * F; cf0 = 1; kd = 0; }                     // private array would be promoted to 8 KB of LDS
        else if (i_ == F || j_ == F) {
            const int r_ = (i_ == F) ? rb : ra, s1 = (i_ == F) ? sb : sa; const bool t2 = (i_ == F) ? b2 : a2;
            id0 = NT_ + r_ - 1; cf0 = s1;
            if (t2) { id1 = NT_ + r_ - 2; cf1 = 1; }
            kd = 1;
        } else {
            id0 = TH(ra, rb); cf0 = sa * sb;
            if (a2) { id1 = TH(ra - 1, rb); cf1 = sb; }
            if (b2) { id2 = TH(ra, rb - 1); cf2 = sa; }
            if (a2 && b2) { id3 = TH(ra - 1, rb - 1); cf3 = 1; }
            kd = 2;
        }
        static_assert(F * (F + 1) / 2 + 2 * F + 2 <= 64, "L.th index fits six bits");
        gdesc[e] = (unsigned)(id0 | ((cf0 & 3) << 6)) | ((unsigned)(id1 | ((cf1 & 3) << 6)) << 8) | ((unsigned)(id2 | ((cf2 & 3) << 6)) << 16) |
                   ((unsigned)(id3 | ((cf3 & 3) << 6)) << 24);
        int sc = 0;                                          // [[I + G11, G1x], [Gx1, Gxx - Sxx]]: the identity and -Sxx
        if (i_ < F && j_ == i_) sc = 1;
        if (i_ > F && j_ > F) { const int r1 = i_ - F, r2 = j_ - F; sc = (r1 == r2) ? (r1 >= 2 ? -2 : -1) : ((r1 - r2 == 1 || r2 - r1 == 1) ? 1 : 0); }
        gkind[e] = (unsigned char)(kd | ((sc & 7) << 2));
    }
    if (!PI) {
        for (int k = threadIdx.x; k <= C; k += WG) { pa_s[PI ? 0 : k] = (R)c.PA[k]; pad_s[k] = c.PA[k]; pa2d_s[k] = c.PA2[k]; if (k < C) a_s[PI ? 0 : k] = (R)c.a[k]; }
    }
    __syncthreads();
    const R* ap = PI ? a_pi[PI ? wv : 0] : a_s;
    const R* pap = PI ? pa_pi[PI ? wv : 0] : pa_s;
    const double* pad = PI ? pad_pi[PI ? wv : 0] : pad_s;
    const double* pa2d = PI ? pa2d_pi[PI ? wv : 0] : pa2d_s;

    // QPs are claimed from a global counter, claim_chunk at a time: iteration counts differ a lot between instances, but one
    // device-wide atomic per QP is a floor of its own (every wavefront of the chip on one address)
    int claim_cur = 0, claim_end = 0;
    const int claim_chunk = max(claim_chunk_, 1);
    // ... and the first static_q / 16 of the launch's QPs are dealt out without any atomic: wavefront g takes [g S, (g + 1) S)
    const int total_qp = order_is_qp ? *count_ptr : 2 * (count_ptr ? *count_ptr : batch);
    const int nwaves = (int)gridDim.x * (WG / 64), gwave = (int)blockIdx.x * (WG / 64) + wv;
    const int share = (int)(((long long)total_qp * static_q) / (16ll * nwaves));
    int st_cur = gwave * share;
    const int st_end = st_cur + share, dyn_base = nwaves * share;
    PH_DECL;
    // The two QPs of an instance (x, y) are neighbours in the hand-out, and what depends on the instance alone is built once: the row
    // mapping (L.k1s / L.w1s) and, with per-instance gait parameters, eta, lambda and its powers, the stability row and its prefix sums
    // (a_pi ... pa2d_pi).  All of it is written in the set-up only, so the second QP finds it in LDS (round 4).
    int last_inst = -1, keep_ovf = 0;
    for (;;) {
        LANE_FRESH();
        PH(0);                                             // 0: between QPs
        int work;
        if (st_cur < st_end) work = st_cur++;
        else {
            if (claim_cur == claim_end) {
                int claimed = 0;
                if (lane == 0) claimed = atomicAdd(work_counter, claim_chunk);
                claim_cur = __builtin_amdgcn_readfirstlane(claimed); claim_end = claim_cur + claim_chunk;
            }
            work = dyn_base + claim_cur++;
        }
        if (work >= total_qp) break;
        const int qpi = order_is_qp ? order[work] : work;
        const int inst = (order && !order_is_qp) ? order[qpi >> 1] : (qpi >> 1), axis = qpi & 1;
        const int qp = 2 * inst + axis;                    // slot of this QP in the working-set history
        const bool same = ISMPC_A_PAIR_SETUP != 0 && inst == last_inst;   // the other axis of the instance this wavefront has just set up
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
        double Qf_d = c.Qf, eta = c.eta, aa_d = c.aa;
        if (PI) {
            const ismpc_a_inst ip = ipar[inst];
            step_ = ip.step; ds_ = ip.ds; Fi = ip.F; plan = ip.plan; Qf_d = ip.Qf;
            if (step_ < 2 || ds_ < 2 || ds_ >= step_ || Fi < 1 || Fi > F || plan < 0 || plan >= c.nplans || !(ip.height > 0) || !(Qf_d > 0)) {
                status |= ISMPC_A_ST_BAD_INDEX; step_ = 2; ds_ = 1; Fi = 1; plan = 0; Qf_d = 1.0; eta = 1.0;
            } else eta = pre[inst].eta;
        }
        // per-instance eta: lambda = exp(-eta dt) and its powers come from the tick prologue (PiPre), one thread per instance
        const double lam_pi = PI ? pre[inst].lam : 0.0, lamC_pi = PI ? pre[inst].lamC : 0.0, lamP_pi = PI ? pre[inst].lamP : 0.0;
        // roots and reciprocals of the gait parameters: formed once per handle (DevA) or per instance (PiPre), not per QP
        const bool dummy_ip = PI && (status & ISMPC_A_ST_BAD_INDEX);           // a rejected record runs on the stand-in values above
        const R sq = dummy_ip ? R(1) : (R)(PI ? pre[inst].sqQf : c.sqQf), isq = dummy_ip ? R(1) : (R)(PI ? pre[inst].isqQf : c.isqQf),
                iQf = dummy_ip ? R(1) : (R)(PI ? pre[inst].iQf : c.iQf);
        const float rstep = dummy_ip ? 0.5f : (PI ? pre[inst].rstep : c.rstep);
        const double inv_dsm1 = PI ? (dummy_ip ? 1.0 : pre[inst].inv_dsm1) : 0.0, inv_ds = dummy_ip ? 1.0 : (PI ? pre[inst].inv_ds : c.inv_ds);
        const double ieta = dummy_ip ? 1.0 : (PI ? pre[inst].ieta : c.ieta);
        const double* fs = PI ? (axis == 0 ? c.plan_x[plan] : c.plan_y[plan]) : (axis == 0 ? c.fsx : c.fsy);
        const double* cl = st.rebuilt ? (axis == 0 ? c.clx1 : c.cly1) : (axis == 0 ? c.clx0 : c.cly0);
        const double cloff = st.rebuilt ? off : 0.0;
        const int ncl = PI ? (c.n_gait - 1) * step_ : c.ncl;
        if (fc < 1 || fc + Fi > c.n_gait || j < 1 || j + P > ncl || j < step_ * (fc - 1) || j > step_ * fc - 1)
            status |= ISMPC_A_ST_BAD_INDEX;
        // the band of EVERY row, relative to the current footstep (the mapping rows sum to one): -(zmp - cur) -+ w/2
        const R zlo = (R)(-(zmp - cur) - c.w / 2), zhi = (R)(-(zmp - cur) + c.w / 2);
        R aa = PI ? (R)pre[inst].aa : (R)aa_d;
        if (PI && !same) {
            // stability row a_i = k1c lambda^i - k2c (quad_walk_no_plots.m:233-238) and its prefix sums for this instance's eta: every
            // row from lambda^i alone (closed forms, PiPre) -- no scan, no reduction, no division
            R* aw = a_pi[PI ? wv : 0]; R* paw = pa_pi[PI ? wv : 0];
            double* padw = pad_pi[PI ? wv : 0]; double* pa2dw = pa2d_pi[PI ? wv : 0];
            const double lam = lam_pi;
            const double k1c = pre[inst].k1c, k2c = pre[inst].k2c, A1 = pre[inst].A1, A2 = pre[inst].A2, B2 = pre[inst].B2, k2c2 = k2c * k2c;
            double lamR = lam;
#pragma unroll
            for (int k = 1; k < RL; ++k) lamR *= lam;
            double lp = ipow6(lamR, lane, nullptr);                       // lambda^(lane RL)
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const int i0 = lane * RL + k;
                const double avk = k1c * lp - k2c;
                lp *= lam;                                                // lambda^(i0 + 1)
                const double om = 1.0 - lp, n1 = (double)(i0 + 1);
                const double run = A1 * om - n1 * k2c, run2 = (A2 * (om * (1.0 + lp)) - B2 * om) + n1 * k2c2;
                if (i0 < C) { aw[i0] = (R)avk; paw[i0 + 1] = (R)run; padw[i0 + 1] = run; pa2dw[i0 + 1] = run2; }
            }
            if (lane == 0) { paw[0] = R(0); padw[0] = 0.0; pa2dw[0] = 0.0; }
            WAVE_LDS_SYNC();
        }

        // ---- per-row data: lane owns ZMP rows lane*RL+1 .. lane*RL+RL (row i = sample i, u index i-1).
        // ks = first mapped footstep (bits 0-3) | row state + 1 (bits 4-5: 0 upper active, 1 free, 2 lower active);
        // pn = previous active row (bits 0-15) | next active row (bits 16-31), kept for every row, active or not
        // One packed int per row: previous active row (bits 0-8) | next active row (9-17) | row state + 1 (18-19: 0 upper active, 1 free,
        // 2 lower active) | first mapped footstep (20-23).  The mapping weight of a row is read back from LDS (L.w1s) where it is used
        // and the row norm of the candidate search is recomputed for violated rows only: u, the multiplier and this int are all a
        // lane keeps per row (round 2 kept six values; the difference is what the four-rows-per-lane shapes spilled).
        R u[RL], mu[RL];
        int pn[RL];
#define W1_(k_)  (L.w1s[lane * RL + (k_)])
#define K1_(k_)  ((pn[k_] >> 20) & 15)
#define STA_(k_) (((pn[k_] >> 18) & 3) - 1)
#define SET_STA_(k_, s_) (pn[k_] = (pn[k_] & ~(3 << 18)) | (((s_) + 1) << 18))
#define PRV_(k_) (pn[k_] & 0x1ff)
#define NXT_(k_) ((pn[k_] >> 9) & 0x1ff)
#define SET_PRV_(k_, p_) (pn[k_] = (pn[k_] & ~0x1ff) | (p_))
#define SET_NXT_(k_, n_) (pn[k_] = (pn[k_] & ~(0x1ff << 9)) | ((n_) << 9))
#define PN_RESET_(k_) (pn[k_] = (pn[k_] & (15 << 20)) | (1 << 18))       /* free, no links; the mapping stays */
#define PN_PRV(v_) ((v_) & 0x1ff)
#define PN_NXT(v_) (((v_) >> 9) & 0x1ff)
        bool ovf = false;
        if (same) {
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const int i = lane * RL + k + 1;
                u[k] = R(0); mu[k] = R(0);
                pn[k] = (i <= C) ? (((int)L.k1s[i - 1] << 20) | (1 << 18)) : (1 << 18);
            }
        } else
#pragma unroll
        for (int k = 0; k < RL; ++k) {
            const int i = lane * RL + k + 1;
            u[k] = R(0); mu[k] = R(0); pn[k] = 1 << 18;
            if (i <= C) {
                int qd = (int)((float)(j + i) * rstep);                          // (j + i) / step_ without the integer-division sequence
                if (qd * step_ > j + i) --qd; else if ((qd + 1) * step_ <= j + i) ++qd;
                int pf = qd - fc + 1; if (pf < 0) pf = 0;
                const int rem = step_ * (fc + pf) - (j + i);
                const R w1k = (rem > ds_) ? R(1) : (R)((double)rem * inv_ds);    // mapping(i, pf+1) = rem / ds; the next column gets 1 - w1
                ovf = ovf || pf > Fi || (rem <= ds_ && pf + 1 > Fi);
                if (pf > 15) pf = 15;
                pn[k] = (pf << 20) | (1 << 18);
                L.k1s[i - 1] = (unsigned char)pf; L.w1s[i - 1] = w1k;
            } else if (i <= WG) L.w1s[i - 1] = R(1);
        }
        if (!same) keep_ovf = __builtin_amdgcn_ballot_w64(ovf) != 0;
        if (keep_ovf) status |= ISMPC_A_ST_OVERFLOW;
        // anticipative tail (quad_walk_no_plots.m:227-231)
        double tail = 0.0;
        if (!(status & ISMPC_A_ST_BAD_INDEX)) {
            if (PI) {
                const double lam = lam_pi;
                double l64;
                double wi = (lamC_pi * lam) * ipow6(lam, lane, &l64) * (1 - lam);      // lambda^(C + 1 + lane) (1 - lambda)
                double tl = 0.0;
#pragma nounroll
                for (int i = C + 1 + lane; i <= P; i += 64) {
                    tl += wi * ((cl_closed(fs, step_, rstep, ds_, inv_dsm1, st.rebuilt != 0, j + i - 1) + cloff) - cur);
                    wi *= l64;
                }
                tail = wave_sum(tl) + lamP_pi * ((cl_closed(fs, step_, rstep, ds_, inv_dsm1, st.rebuilt != 0, P - 1) + cloff) - cur);
            } else {
                // the handle's own parameters: the sum over the samples depends on the tick index alone (DevA::tlx0 ...: built by the host)
                const double* tt = st.rebuilt ? (axis == 0 ? c.tlx1 : c.tly1) : (axis == 0 ? c.tlx0 : c.tly0);
                tail = tt[j] + (cloff - cur) * c.sumw;
            }
        }
        last_inst = (status & ISMPC_A_ST_BAD_INDEX) ? -1 : inst;
        const R beq = (R)(pos + vel * ieta - zmp - tail);
        // ---- kinematic row r and footstep f_r (relative to the current one) live in lane r (1..F); Khat_r = sqrt(Qf) (f_r - f_{r-1})
        R fr = R(0), klo = R(-INFINITY), khi = R(INFINITY), muK = R(0);
        int kact = 0;
        if (klane) {
            const int r = lane;
            double bup = axis == 0 ? c.disp_forw : (c.disp_L / 2 + c.disp_L / 2);
            if (fc == 1 && r == 1) bup = axis == 0 ? c.disp_forw_dummy : (c.disp_L / 2 + c.disp_L / 2);
            khi = (R)bup; klo = (R)(-bup);                                       // f_1 - cur, f_r - f_{r-1}: symmetric in these coordinates
            // (a lane beyond this instance's footstep count would read past the plan near its end: no load)
            fr = ((status & ISMPC_A_ST_BAD_INDEX) || (PI && r > Fi)) ? R(0) : (R)((fs[fc + r - 1] + off) - cur);
            if (PI && r > Fi) { khi = R(INFINITY); klo = R(-INFINITY); fr = R(0); }   // beyond this instance's horizon: no variable, no row
        }
        const R knrm = (lane >= 2) ? sq * R(0.70710678118654752440) : sq;   // 1 / |K_r|_{H^-1}: |kvec_r|^2 = 2 (r >= 2) or 1
        int iters = 0, qz = 0, qk = 0;
        bool defer_qp = false;                                // fp32 solve: handed to the fp64 re-solve launch (see the block-solve check)
#ifdef ISMPC_A_DIAG
        int dg_ns = 0, dg_cold = 0, dg_q0 = 0, dg_part = 0, dg_why = 0;   // diagnostic build (scripts/iters_hist.py): see the packing at the output
#endif
        R muE = R(0);
        PH(1);                                             // 1: record, gait parameters, per-row set-up, tail
        bool done_opt = false;                                // the block passes ended on a checked optimum: nothing left to do
        if (status == 0) {
            // ---- equality first: u = (b / a'a) a
            const R t0 = beq / aa;
#pragma unroll
            for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; u[k] = (i <= C) ? t0 * ap[i - 1] : R(0); }
            muE = t0;
            for (int e = lane; e < m * m; e += 64) L.G[e] = G_CONST(gkind[e]);     // no active ZMP row: G = 0, the constant part stays
            if (lane <= F + 1) L.pf[lane] = fr;
            WAVE_LDS_SYNC();

            // row values at the current point: v_i = dt cumsum(u)_i - M_i f, in vv[]; leaves f in L.fl
            auto row_values = [&](R (&vv)[RL]) __attribute__((always_inline)) {
                LANE_FRESH();
                if (lane <= F + 1) L.fl[lane] = fr;
                WAVE_LDS_SYNC();
                R lc = R(0);
#pragma unroll
                for (int k = 0; k < RL; ++k) { lc += u[k]; vv[k] = lc; }
                const R bs = wave_scan_up(lc) - lc;
#pragma unroll
                for (int k = 0; k < RL; ++k) { const R w1k = W1_(k); vv[k] = dt * (vv[k] + bs) - (w1k * L.fl[K1_(k)] + (R(1) - w1k) * L.fl[K1_(k) + 1]); }
                PH(2);                                     // 2: row values
            };

            // D = a'a - G_EE >= 0, the (stability, stability) entry of the small system with its sign flipped.  Both terms are sums
            // over the same samples -- G_EE = sum over gaps (p, i] between consecutive active rows of (sum a)^2 / (i - p) -- and
            // they cancel completely when every ZMP row is active (the stability row then lies in the span of the active rows
            // but for the footstep coupling, which is 1/Qf small): formed as ONE sum of per-gap terms
            //     [sum a^2 - (sum a)^2 / (i - p)]  +  (sum of a^2 past the last active row),
            // each from the fp64 prefix sums, it is exact where it matters (a gap of one row contributes exactly 0).
            auto stability_defect = [&]() __attribute__((always_inline)) -> double {
                double dl = 0.0;
#pragma unroll
                for (int k = 0; k < RL; ++k) {
                    const int i = lane * RL + k + 1;
                    if (i <= C && STA_(k) != 0) {
                        const int pv = PRV_(k);
                        const double sa = pad[i] - pad[pv];
                        dl += (pa2d[i] - pa2d[pv]) - sa * sa * frcp((double)(i - pv));
                        if (NXT_(k) == 0) dl += pa2d[C] - pa2d[i];
                    }
                }
                if (qz == 0 && lane == 0) dl = pa2d[C];
                return wave_sum(dl);
            };
            // the same terms one at a time, for the rows that enter / leave in the Goldfarb-Idnani loop (wave-uniform, fp64)
            auto gap_term = [&](int p_, int i_) __attribute__((always_inline)) -> double {
                const double sa = pad[i_] - pad[p_];
                return (pa2d[i_] - pa2d[p_]) - sa * sa * frcp((double)(i_ - p_));
            };
            auto tail_term = [&](int p_) __attribute__((always_inline)) -> double { return pa2d[C] - pa2d[p_]; };
            double Dd = pa2d[C];
            // mean of a over the samples p+1 .. j (consecutive active rows p < j), from the fp64 prefix sums
            auto abar = [&](int p_, int j_) __attribute__((always_inline)) -> R { return (R)(pad[j_] - pad[p_]) * rinv[j_ - p_]; };
            R Dee = aa;

            // ---- small quasi-definite system  [[I+G11, G1x],[Gx1, Gxx - Sxx]] cc = rhs (L.hx), G in L.G; unknown order:
            // 0..F-1 footstep columns, F = stability row, F+1..2F = Khat_1..F (rows outside kmask: pinned to 0).  Returns
            // this lane's cc[lane] and leaves cc in L.cc.
            // NOK (a compile-time tag): no kinematic row is in the working set -- always true in the block passes, usually in the exact
            // steps.  The pinned unknowns F+1..2F then decouple (their rows and columns are unit vectors, their right-hand sides 0): the
            // elimination runs on the leading (F+1) x (F+1) block only -- (F+1)(F+2)/2 row updates instead of (F+1)(3F+2)/2, and no loads
            // of the pinned columns -- and gives the same numbers (the dropped updates multiply exact zeros).
            auto solve_small = [&](auto NOK, const unsigned long long kmask) __attribute__((always_inline)) -> R {
                LANE_FRESH();
                constexpr bool nok = decltype(NOK)::value;
                constexpr int mu_ = nok ? F + 1 : m;                                // unknowns the elimination touches
                // lane i < m owns row i of the augmented matrix in registers; the pivot row travels by readlane: no LDS
                // traffic and no barriers inside the elimination
                const int i = lane < m ? lane : m - 1;
                const bool ipin = i > F && (nok || !((kmask >> (i - F)) & 1ull));
                R Tr[mu_ + 1];
#pragma unroll
                for (int jj = 0; jj < mu_; ++jj) {
                    R val = L.G[i * m + jj];                                        // G plus the constant part (identity, -Sxx): see gkind
                    if (jj == F && i == F) val = -Dee;                              // G_EE - a'a without the cancellation
                    const bool jpin = !nok && jj > F && !((kmask >> (jj - F)) & 1ull);
                    if (ipin || jpin) val = (i == jj) ? R(-1) : R(0);
                    Tr[jj] = val;
                }
                Tr[mu_] = ipin ? R(0) : L.hx[i];
#pragma unroll
                for (int kk = 0; kk < mu_; ++kk) {                               // Gauss-Jordan, no pivoting (quasi-definite)
                    if (!nok && kk > F && !((kmask >> (kk - F)) & 1ull)) continue;   // pinned unknown: its column is already e_kk
                    if (PI && kk < F && kk >= Fi) continue;                      // footstep beyond this instance's horizon: no row maps to
                                                                                 // it, its row and column of G are zero, the pivot is 1
                    const R ipv = frcp(rl(Tr[kk], kk));
                    const R fct = (lane == kk) ? R(0) : Tr[kk] * ipv;
#pragma unroll
                    for (int jj = kk + 1; jj <= mu_; ++jj) Tr[jj] -= fct * rl(Tr[jj], kk);
                }
                R dg = nok ? R(-1) : Tr[0];                                      // (a pinned row of the reduced form: diagonal -1, right-hand side 0)
#pragma unroll
                for (int jj = (nok ? 0 : 1); jj < mu_; ++jj) if (lane == jj) dg = Tr[jj];
                const R cc_e = (lane < m) ? Tr[mu_] * frcp(dg) : R(0);           // lane e: cc[e]
                if (lane < m) L.cc[lane] = cc_e;
                WAVE_LDS_SYNC();
                PH(3);                                     // 3: small system (Gauss-Jordan)
                return cc_e;
            };

            // ---- one structured solve for a whole working set (the ZMP rows by their state bits, the kinematic rows in kmask / kact):
            // minimiser u, f and all multipliers; leaves G(W) in L.G and prv / nxt of every row
            auto block_solve = [&](const unsigned long long kmask) __attribute__((always_inline)) {
                LANE_FRESH();
                PH(10);                                    // 10: block passes: selection (what ran since the row values)
                // ---- previous / next active row of every row (active or not): exclusive max scan, exclusive suffix min scan
                int nact = 0;
                R cvr[RL];
                // active kinematic rows: right-hand sides sqrt(Qf) (bound_r - (p_r - p_{r-1})) travel to the lanes of their unknowns
                if (klane) L.d1[lane - 1] = (kact != 0) ? sq * ((kact > 0 ? klo : khi) - (L.pf[lane] - L.pf[lane - 1])) : R(0);
                {
                    int lmax = 0, lmin = 1 << 30;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        const int sk = STA_(k);
                        const bool act = i <= C && sk != 0;
                        if (act) { lmax = max(lmax, i); lmin = min(lmin, i); }
                        nact += __builtin_popcountll(__builtin_amdgcn_ballot_w64(act));
                        // c_i = bound_i + M_i . plan footsteps
                        const R w1c = W1_(k);
                        cvr[k] = act ? (sk > 0 ? zlo : zhi) + (w1c * L.pf[K1_(k)] + (R(1) - w1c) * L.pf[K1_(k) + 1]) : R(0);
                        if (i <= C) L.sv[i - 1] = cvr[k];                    // c of every row, for its successor
                    }
                    int run = dpp_i<0x138, 0xf>(0, wave_scan_max_i(lmax));
#pragma unroll
                    for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; SET_PRV_(k, run); if (i <= C && STA_(k) != 0) run = i; }
                    const int rev = __shfl(lmin, 63 - lane);
                    const int ex = dpp_i<0x138, 0xf>(1 << 30, -wave_scan_max_i(-rev));
                    const int nx = __shfl(ex, 63 - lane);
                    run = (nx == (1 << 30)) ? 0 : nx;
#pragma unroll
                    for (int k = RL - 1; k >= 0; --k) { const int i = lane * RL + k + 1; SET_NXT_(k, run); if (i <= C && STA_(k) != 0) run = i; }
                }
                WAVE_LDS_SYNC();
                PH(4);                                     // 4: block solve: links (prev / next scans)
                // ---- G = V'K^-1 V / dt^2 and g = V'K^-1 c / dt^2 as sums over consecutive active pairs (p, i) of
                // d d' / gap, d = V_i - V_p.  V_i = Phi(theta_i) + dt PA_i e_E with theta_i the row's mapping weights over the
                // F footstep columns and Phi a fixed sparse map, so everything follows from the Gram sums of
                // [dtheta (F) | dt dPA | dc] weighted by 1 / (dt^2 gap): each lane adds its own rows, then the sums are folded
                // over the wavefront (fold_step: about one exchange-add per sum instead of six).
                {
                    constexpr int NT = F * (F + 1) / 2, NR = 2 * F + 2;
                    auto pair_diffs = [&](int k, R (&dth)[F], R& om, R& dE, R& dc) __attribute__((always_inline)) {
                        const int i = lane * RL + k + 1;
                        const int p_ = PRV_(k);
                        int pk1 = -8; R pw1 = R(0), ppa = R(0), pc = R(0);                   // V_0 = 0, c_0 = 0
                        if (p_ > 0) { pk1 = L.k1s[p_ - 1]; pw1 = L.w1s[p_ - 1]; ppa = pap[p_]; pc = L.sv[p_ - 1]; }
                        const R w1k = W1_(k);
                        const R pw2 = (p_ > 0) ? R(1) - pw1 : R(0), w2 = R(1) - w1k;
                        om = idt2 * rinv[i - p_];
                        dE = dt * (pap[i] - ppa); dc = cvr[k] - pc;
                        const int k1k = K1_(k);
#pragma unroll
                        for (int r = 1; r <= F; ++r) {
                            const R ti = (r == k1k) ? w1k : ((r == k1k + 1) ? w2 : R(0));
                            const R tp = (r == pk1) ? pw1 : ((r == pk1 + 1) ? pw2 : R(0));
                            dth[r - 1] = ti - tp;
                        }
                    };
                    {
                        // The NS = (F+2)(F+3)/2 - 1 sums are taken in passes of at most GCAP accumulators (round 4): all of them at once is 20
                        // live registers at F = 4 and 35 at F = 6 on top of a lane's rows, which is what the three- and four-rows-per-lane shapes
                        // spilled to scratch (<double,3,4,false> 14 registers, <float,4,6,true> 32: 7-106x the algorithmic HBM traffic).  A pass
                        // re-forms the pair differences of the lane's rows (a few loads from LDS) and keeps only its own products; the
                        // decomposition into power-of-two chunks that the fold works on is the one wave_fold_sums has always used up to F = 5
                        // (bit-identical sums), at F = 6 the leading chunk of 32 becomes two of 16.
                        constexpr int NS = NT + NR;
                        auto gram = [&](auto self, auto S_) __attribute__((always_inline)) -> void {
                            constexpr int S = decltype(S_)::value;
                            if constexpr (S < NS) {
                                constexpr int E = (NS - S >= GCAP) ? S + GCAP : NS, NP = E - S;
                                R acc[NP];
#pragma unroll
                                for (int t = 0; t < NP; ++t) acc[t] = R(0);
#pragma unroll
                                for (int k = 0; k < RL; ++k) {
                                    const int i = lane * RL + k + 1;
                                    if (i <= C && STA_(k) != 0) {
                                        R dth[F], om, dE, dc;
                                        pair_diffs(k, dth, om, dE, dc);
                                        int t = 0;
#pragma unroll
                                        for (int r = 0; r < F; ++r) {
                                            const R od = om * dth[r];
#pragma unroll
#define G_IN(t_) ((t_) >= S && (t_) < E)
#define G_AT(t_) acc[G_IN(t_) ? (t_) - S : 0]           /* (the index of a sum that is not this pass's is never used: no out-of-range constant) */
                                            for (int q = r; q < F; ++q) { if (G_IN(t)) G_AT(t) += od * dth[q]; ++t; }
                                            if (G_IN(NT + r)) G_AT(NT + r) += od * dE;
                                            if (G_IN(NT + F + r)) G_AT(NT + F + r) += od * dc;
                                        }
                                        if (G_IN(NT + 2 * F)) G_AT(NT + 2 * F) += om * dE * dE;
                                        if (G_IN(NT + 2 * F + 1)) G_AT(NT + 2 * F + 1) += om * dE * dc;
#undef G_IN
#undef G_AT
                                    }
#ifdef ISMPC_A_GRAM_SCHED
                                    __builtin_amdgcn_sched_barrier(0);       // one row's pair differences live at a time
#endif
                                }
                                if constexpr (NP == GCAP) {
                                    const R part = fold_pow2<R, GCAP, 1>(acc, lane);
                                    const R tot = finish_sum<R, GCAP>(part);
                                    if (lane < GCAP) L.th[S + lane] = tot;
                                } else wave_fold_sums<R, NP>(acc, L.th + S, lane);
                                self(self, std::integral_constant<int, E>{});
                            }
                        };
                        gram(gram, std::integral_constant<int, 0>{});
                        PH(5);                             // 5: block solve: Gram accumulation (and, since round 4, the folds)
                    }
                    WAVE_LDS_SYNC();
                    PH(6);                                 // 6: block solve: fold over the wavefront
                    // G from the Gram sums through the per-workgroup table (gidx / gcoef / gkind, built at kernel start)
                    for (int e = lane; e < m * m; e += 64) {
                        const unsigned dsc = gdesc[e];
                        R val = R(0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) val += (R)((int)(dsc << (24 - 8 * t)) >> 30) * L.th[(dsc >> (8 * t)) & 63u];
                        const int kd = gkind[e];
                        L.G[e] = val * ((kd & 3) == 0 ? R(1) : ((kd & 3) == 1 ? isq : isq * isq)) + G_CONST(kd);
                    }
                    if (lane < m) {
                        R gv;
                        if (lane == F) gv = L.th[NT + 2 * F + 1] - beq;
                        else {
                            const int ra = lane < F ? lane + 1 : lane - F;
                            gv = (lane < F ? R(1) : R(-1)) * L.th[NT + F + ra - 1];
                            if (lane > F && ra >= 2) gv += L.th[NT + F + ra - 2];
                            gv *= isq;
                        }
                        if (lane > F) gv -= L.d1[lane - F - 1];
                        L.hx[lane] = gv;
                    }
                }
                WAVE_LDS_SYNC();
                qz = nact;
                PH(7);                                     // 7: block solve: G and right-hand side from the sums
                Dd = stability_defect(); Dee = (R)Dd;
                PH(8);                                     // 8: block solve: stability defect
                if (kmask == 0ull) (void)solve_small(std::true_type{}, 0ull); else (void)solve_small(std::false_type{}, kmask);
                const R cEw = L.cc[F];
                // comb[r] = (cc[r-1] - ck[r] + ck[r+1]) / sqrt(Qf), r = 1..F: what a row sees through its two footstep columns
                // (ck = the kinematic unknowns, 0 where pinned)
                if (lane <= F + 1) L.comb[lane] = klane ? (L.cc[lane - 1] - L.cc[F + lane] + (lane + 1 <= F ? L.cc[F + lane + 1] : R(0))) * isq : R(0);
                if (klane) muK = (kact != 0) ? (kact > 0 ? R(1) : R(-1)) * L.cc[F + lane] : R(0);
                WAVE_LDS_SYNC();
                R sl[RL];                                                    // s~_i = c_i - M~_i . cc on the active rows (the stability
#pragma unroll                                                               // column dt PA_i cE is carried separately, see below)
                for (int k = 0; k < RL; ++k) {
                    const int i = lane * RL + k + 1;
                    sl[k] = R(0);
                    if (i <= C && STA_(k) != 0)
                        { const R w1k = W1_(k); sl[k] = cvr[k] - (w1k * L.comb[K1_(k)] + (R(1) - w1k) * L.comb[K1_(k) + 1]); }
                    if (i <= C) L.sv[i - 1] = sl[k];
                }
                WAVE_LDS_SYNC();
                // multipliers (tridiagonal K^-1: second differences of s over the active rows) and u = dt suffix(lambda) + lambda_E a.
                // The suffix sum telescopes: between two consecutive active rows p < j, u is the SLOPE of s over the gap (zero past the
                // last active row) -- no scan, and a first difference instead of summed second differences.  With
                // s = s~ - dt PA cE the slope is (s~_j - s~_p) / ((j - p) dt) - cE mean(a over the gap), so
                //     u_i = slope~ + cE (a_i - mean a):
                // in a run of consecutive active rows the stability multiplier (huge when the whole horizon is active) drops out
                // exactly, as it must -- u is then fixed by the rows alone.
#pragma unroll
                for (int k = 0; k < RL; ++k) {
                    const int i = lane * RL + k + 1;
                    const int sk = STA_(k);
                    const int pv = PRV_(k), nx = NXT_(k);
                    const int jj = (sk != 0) ? i : nx;                           // first active row at or after this one (0: none)
                    R r_ = R(0), uu = R(0);
                    if (i <= C) {
                        if (jj > 0) {
                            const R sp = pv > 0 ? L.sv[pv - 1] : R(0);
                            const R sj = (sk != 0) ? sl[k] : L.sv[jj - 1];
                            const R d1 = (sj - sp) * rinv[jj - pv];
                            const R ab1 = abar(pv, jj);
                            uu = d1 * idt + cEw * (ap[i - 1] - ab1);
                            if (sk != 0) {
                                r_ = d1 * idt2 - idt * cEw * ab1;
                                if (nx > 0) r_ -= (L.sv[nx - 1] - sl[k]) * rinv[nx - i] * idt2 - idt * cEw * abar(i, nx);
                            }
                        } else uu = cEw * ap[i - 1];
                    }
                    mu[k] = sk > 0 ? r_ : -r_;
                    u[k] = uu;
                }
                if (klane) fr = L.pf[lane] - L.comb[lane];
                muE = cEw;
                WAVE_LDS_SYNC();
                PH(9);                                     // 9: block solve: slopes, multipliers, u
            };

            // ================= block warm start (primal-dual active-set passes) =================
            // The loop below adds one row per iteration and a nominal tick ends with 40-70 active rows.  Before it, up to
            // c.warm_add passes put every violated ZMP row into the working set at once (and take out rows whose multiplier
            // is not positive), each followed by ONE structured solve for the whole set: G = V'K^-1 V and g = V'K^-1 c from
            // one sweep over the active rows (K^-1 is tridiagonal: gaps only), the (F+1)-unknown system, a tridiagonal apply
            // and a suffix sum.  Up to c.warm_drop more passes only remove rows with negative multipliers.  What is left is a
            // valid starting pair for Goldfarb-Idnani (minimiser on its working set, multipliers >= 0), which finishes the
            // job and owns the kinematic rows; if the passes do not get there the solve starts cold.  Same optimum either way.
            // closed loop: the working set this instance ended the previous tick with is the first guess of the block passes
            bool have_guess = false;
            if (c.warm_add > 0 && hist != nullptr && hist_load) {
                const unsigned long long* hq = hist + (size_t)qp * 8;
                unsigned long long any_ = 0ull;
#pragma unroll
                for (int k = 0; k < RL; ++k) any_ |= hq[k] | hq[4 + k];           // only the words the store below writes
                have_guess = any_ != 0ull;
            }
            // Three phases share one Goldfarb-Idnani loop.  phase 0 (one-shot ticks): up to c.warm_gi rows enter one at a time
            // from the equality-only point -- its violated stretch is a poor predictor of the optimal working set (a long
            // violated run often ends as one or two touching points), the stretch left after two or three exact steps is a
            // good one; phase 1: the block passes, from wherever phase 0 stopped (or from the previous tick's working set);
            // phase 2: Goldfarb-Idnani to the end.  Between 1 and 2, up to c.warm_rounds rounds of (c.warm_round_adds more
            // Goldfarb-Idnani additions, then the passes again): when the passes collapse -- an over-constrained intermediate
            // set turns most multipliers negative at once and the drop-only passes strip the set to a row or two -- the exact
            // steps re-seed them instead of re-adding a hundred rows one at a time.
            int phase = (c.warm_add <= 0) ? 2 : ((c.warm_gi > 0 && !have_guess) ? 0 : 1);
            int gi_limit = c.warm_gi, rounds_left = c.warm_rounds;
            int pass_solves = 0; bool pass_cold = false;          // what the last run of the passes did
            for (;;) {
            if (phase == 1) {
                phase = 2;
                bool cold = false, force_add = false;
                int extra = c.warm_extra, nsolve = 0;
                // the previous tick's working set, moved down by one row (the horizon advanced by one sample); any guess is
                // safe, the passes validate it
                int guess[RL];
#pragma unroll
                for (int k = 0; k < RL; ++k) guess[k] = 0;
                if (have_guess) {
                    const unsigned long long* hq = hist + (size_t)qp * 8;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const unsigned long long lo_ = (k < RL - 1) ? hq[k + 1] : (hq[0] >> 1);          // rows on the lower bound
                        const unsigned long long hi_ = (k < RL - 1) ? hq[4 + k + 1] : (hq[4] >> 1);      // rows on the upper bound
                        const int i = lane * RL + k + 1;
                        if (i <= C) guess[k] = ((lo_ >> lane) & 1ull) ? 1 : (((hi_ >> lane) & 1ull) ? -1 : 0);
                    }
                }
                for (int pass = 0; ; ++pass) {
                    LANE_FRESH();
                    const bool adding = pass < c.warm_add || force_add;
                    force_add = false;
                    // ---- row values at the current point; the new working set
                    R vv[RL];
                    row_values(vv);
                    // ---- rows that leave: multiplier not positive (while adding) / negative (drop-only passes).  Such a row
                    // usually sits at the end of a run of consecutive rows on the same bound, and the run has to shrink by
                    // more than that one row.  How far: a multiplier at row i acts on u_j, j <= i, exactly like one at any later
                    // row, so the multipliers of the rows cut off a run's end, lumped onto the new end row, leave the earlier
                    // horizon as it is -- and the new end must come out positive.  A negative end therefore takes with it the
                    // rows whose multipliers, summed from that end, are still <= 0 (docs/models/proto_passes.py: 15 % less work than
                    // doubling the cut from pass to pass, worst QP 22 -> 11-18 units, no state between passes).  Never the row
                    // at the other end: a run shrinks to one row, which leaves only on its own multiplier (cutting a two-row
                    // touching point away un-pins the trajectory there, every other multiplier turns negative at once and the
                    // next pass starts from nothing).
                    bool xdrop[RL], negr[RL], anyneg = false;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        negr[k] = STA_(k) != 0 && (adding ? !(mu[k] > R(0)) : (mu[k] < R(0)));
                        xdrop[k] = false; anyneg = anyneg || negr[k];
                    }
                    const bool wave_neg = __builtin_amdgcn_ballot_w64(anyneg) != 0;
                    if (have_guess && nsolve == 1 && wave_neg) {
                        // the previous tick's working set, solved as it came: if most of its multipliers are negative the set is
                        // over-full (a horizon that was pinned end to end and is being released), every one of them would be
                        // dropped and the passes would start from nothing -- solve this QP the one-shot way instead
                        int nneg = 0, nall = 0;
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            nneg += __builtin_popcountll(__builtin_amdgcn_ballot_w64(negr[k]));
                            nall += __builtin_popcountll(__builtin_amdgcn_ballot_w64(STA_(k) != 0));
                        }
                        if (2 * nneg > nall) {
#ifdef ISMPC_A_DIAG
                            dg_why = 3;
#endif
                            cold = true; break;
                        }
                    }
                    if (wave_neg && c.warm_peel_end != 0) {
                        const int sprev = dpp_i<0x138, 0xf>(0, STA_(RL - 1)), snext = dpp_i<0x130, 0xf>(0, STA_(0));
                        int lst = 0, len_ = 1 << 30;                            // this lane's last run start / first run end
                        bool isst[RL], isen[RL];
                        // inclusive prefix sums P of the (signed) multipliers over the active rows, each with its row's verdict in the
                        // last mantissa bit: the sum over any stretch of a run is a difference of two of them
                        R Pk[RL], lc = R(0);
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int i = lane * RL + k + 1;
                            const int sk = STA_(k);
                            const int sb = k > 0 ? STA_(k > 0 ? k - 1 : 0) : sprev, sa = k < RL - 1 ? STA_(k < RL - 1 ? k + 1 : 0) : snext;
                            isst[k] = sk != 0 && sb != sk; isen[k] = sk != 0 && sa != sk;
                            if (isst[k]) lst = i;
                            if (isen[k]) len_ = min(len_, i);
                            if (sk != 0) lc += mu[k];
                            Pk[k] = lc;
                        }
                        const R bs = wave_scan_up(lc) - lc;
#pragma unroll
                        for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; Pk[k] += bs; if (i <= C) L.sv[i - 1] = with_flag(Pk[k], negr[k]); }
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
                            if (STA_(k) != 0 && runlo[k] >= 1 && runhi[k] <= C && runlo[k] != runhi[k]) {
                                const R eh = L.sv[runhi[k] - 1], el = L.sv[runlo[k] - 1];
                                const R pb = runlo[k] > 1 ? L.sv[runlo[k] - 2] : R(0);                      // P just before the run
                                if (flag_of(eh) && i > runlo[k] && !((eh - Pk[k]) + mu[k] > R(0))) xdrop[k] = true;
                                if (flag_of(el) && i < runhi[k] && !(Pk[k] - pb > R(0))) xdrop[k] = true;
                            }
                        }
                        WAVE_LDS_SYNC();
                    }
                    bool changed = false, off_bound = false;
                    R aul = R(0);
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        if (i <= C) {
                            const int os = STA_(k);
                            int ns = os;
                            const R v = vv[k];
                            aul += ap[i - 1] * u[k];
                            if (have_guess && pass == 0) ns = guess[k];
                            else if (ns != 0) {
                                // an active row must sit on its bound after the block solve; if it does not, the solve broke down
                                const R bd = ns > 0 ? zlo : zhi;
                                off_bound = off_bound || !(fabs(v - bd) <= (R)NM::bound_rel * (fabs(v) + fabs(bd)) + (R)NM::bound_abs);
                                if (negr[k] || xdrop[k]) ns = 0;
                            } else if (adding) {
                                const R tol = (R)NM::viol_rel * (fabs(v) + fmax(fabs(zlo), fabs(zhi))) + (R)NM::viol_abs;
                                if (v - zlo < -tol) ns = 1; else if (zhi - v < -tol) ns = -1;
                            }
                            changed = changed || ns != os;
                            SET_STA_(k, ns);
                        }
                    }
                    if (nsolve > 0) {
                        const R eqr = wave_sum(aul) - beq;                         // ... and the stability row must hold
                        if (__builtin_amdgcn_ballot_w64(off_bound) != 0 || !(fabs(eqr) <= (R)NM::eq_rel * (R(1) + fabs(beq)))) {
#ifdef ISMPC_A_DIAG
                            dg_why = 1;
#endif
                            // fp32: a working set that pins (nearly) the whole horizon is beyond the block solve's accuracy, and the
                            // cold start that follows is 100-150 one-row steps -- one such QP in 30 000 makes its launch 2-3x longer.
                            // It goes to the fp64 instantiation instead (a small launch right behind this one)
                            if (sizeof(R) == 4 && defer_list != nullptr) defer_qp = true;
                            cold = true; break;
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(changed) == 0) {               // a valid pair (and, while adding, nothing violated)
                        if (!adding && extra > 0) {
                            // valid after drop-only passes.  Goldfarb-Idnani would now take the violated rows one at a time; when
                            // many are left, one more adding pass (all of them at once) is the cheaper way on
                            int nv = 0;
#pragma unroll
                            for (int k = 0; k < RL; ++k) {
                                const int i = lane * RL + k + 1;
                                const R v = vv[k];
                                const R tol = (R)NM::viol_rel * (fabs(v) + fmax(fabs(zlo), fabs(zhi))) + (R)NM::viol_abs;
                                nv += __builtin_popcountll(__builtin_amdgcn_ballot_w64(i <= C && STA_(k) == 0 && (v - zlo < -tol || zhi - v < -tol)));
                            }
                            if (nv >= c.warm_min_viol) { --extra; force_add = true; continue; }
                        }
                        if (adding && nsolve > 0) {
                            // every ZMP row was just evaluated at this point (none violated, active ones on their bounds, the
                            // stability row holds, multipliers positive); with the kinematic rows inside their limits this
                            // IS the optimum: skip the Goldfarb-Idnani search and the final re-check
                            const R fprev = dppv<0x111, 0xf, true>(R(0), fr);
                            bool kbad = false;
                            if (klane && khi < R(INFINITY)) {
                                const R vk = fr - fprev, tol = (R)NM::viol_rel * (fabs(vk) + fmax(fabs(klo), fabs(khi))) + (R)NM::viol_abs;
                                kbad = !(vk - klo >= -tol && khi - vk >= -tol);
                            }
                            done_opt = __builtin_amdgcn_ballot_w64(kbad) == 0;
                        }
                        break;
                    }
                    if (nsolve >= c.warm_add + c.warm_drop + c.warm_extra * (1 + c.warm_drop)) {                        // budget spent: start cold
#ifdef ISMPC_A_DIAG
                        dg_why = 2;
#endif
                        cold = true; break;
                    }
                    ++nsolve; ++iters;
                    block_solve(0ull);                                           // kinematic rows stay out of the block phase (tried: the
                                                                                 // passes adding / dropping them too changes nothing on the bench workloads)
                }
#ifdef ISMPC_A_DIAG
                dg_ns += nsolve; dg_cold = cold ? dg_why : 0; dg_q0 = cold ? 0 : qz;
#endif
                PH(10);
                pass_solves = nsolve; pass_cold = cold;
                if (defer_qp) break;
                if (cold) {
#pragma unroll
                    for (int k = 0; k < RL; ++k) { const int i = lane * RL + k + 1; PN_RESET_(k); mu[k] = R(0); u[k] = (i <= C) ? t0 * ap[i - 1] : R(0); }
                    if (klane) fr = L.pf[lane];
                    muE = t0; qz = 0; Dd = pa2d[C]; Dee = (R)Dd;
                    for (int e = lane; e < m * m; e += 64) L.G[e] = G_CONST(gkind[e]);
                    WAVE_LDS_SYNC();
                    // a previous-tick guess that the passes could not repair: solve this QP the way a one-shot tick is solved
                    // (a closed loop's tick is as long as its slowest QP, and a cold Goldfarb-Idnani solve is 50-100 steps)
                    if (have_guess && c.warm_gi > 0) { phase = 0; gi_limit = c.warm_gi; }
                }
                have_guess = false;
            }
            if (done_opt) break;
            if (phase == 2 && rounds_left > 0 && pass_solves > 0 && !pass_cold) { phase = 0; gi_limit = c.warm_round_adds; --rounds_left; }

            int gi_adds = 0;
            bool gi_leave = false;                                // feasible, or failed: nothing more to do
            for (;;) {
                // the exact steps of phase 0 are spent: the block passes take over -- and evaluate every row themselves, so no search for a row
                // that would not be added (round 4: that search was ~170 vector instructions of every one-shot solve, for a `break`)
                if (phase == 0 && gi_adds >= gi_limit) break;
                // ================= most violated inactive row =================
                LANE_FRESH();
                R cand = R(0), craw = R(0); int code = 0;
                {
                    R vv[RL];
                    row_values(vv);
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        if (i <= C && STA_(k) == 0) {
                            const R v = vv[k];
                            const R vl = v - zlo, vh = zhi - v;
                            const R tol = (R)NM::viol_rel * (fabs(v) + fmax(fabs(zlo), fabs(zhi))) + (R)NM::viol_abs;
                            if (vl < -tol || vh < -tol) {
                                // 1 / |row_i|_{H^-1} (float: it only ranks candidates): |row|^2 = dt^2 i + |M_i|^2 / Qf over the footstep columns
                                const float w1f = (float)W1_(k), w2f = 1.0f - w1f;
                                const R nr = (R)rsq_f((float)(c.dt * c.dt) * (float)i + (w2f * w2f + (K1_(k) >= 1 ? w1f * w1f : 0.0f)) * (float)iQf);
                                if (vl < -tol && vl * nr < cand) { cand = vl * nr; craw = vl; code = 2 * i; }
                                if (vh < -tol && vh * nr < cand) { cand = vh * nr; craw = vh; code = 2 * i + 1; }
                            }
                        }
                    }
                    const R fprev = dppv<0x111, 0xf, true>(R(0), fr);             // f_{r-1} (lane 0 holds f_0 = 0)
                    if (klane && kact == 0) {
                        const R v = fr - fprev;
                        const R vl = v - klo, vh = khi - v;
                        const R tol = (R)NM::viol_rel * (fabs(v) + fmax(fabs(klo), fabs(khi))) + (R)NM::viol_abs;
                        if (vl < -tol && vl * knrm < cand) { cand = vl * knrm; craw = vl * sq; code = 2 * (C + lane); }
                        if (vh < -tol && vh * knrm < cand) { cand = vh * knrm; craw = vh * sq; code = 2 * (C + lane) + 1; }
                    }
                }
                const R vmin = wave_min(cand);
                PH(11);                                    // 11: Goldfarb-Idnani: search for the most violated row
                if (!(vmin < R(0))) { gi_leave = true; break; }                   // feasible: done
                const int wl = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(cand == vmin));
                const int cd = rl(code, wl);
                R sviol = rl(craw, wl);
                const int row = cd >> 1;
                const R sg = (cd & 1) ? R(-1) : R(1);
                const bool isZ = row <= C;
                if (phase == 0 && (gi_adds >= gi_limit || !isZ)) break;          // hand over to the block passes (which leave the kinematic rows alone)
                const int kr = row - C;                                           // kinematic index when !isZ
                // ---- the new row: border row Vp (one element per lane), footstep part mt, norm, border products dX
                int p_k1 = 0; R p_w1 = R(1), p_pa = R(0);
                if (isZ) { p_k1 = L.k1s[row - 1]; p_w1 = L.w1s[row - 1]; p_pa = pap[row]; }
                const R p_w2 = R(1) - p_w1;
                const R vp = (isZ && lane < m) ? border_elem<R, F>(lane, p_k1, p_w1, p_pa, dt, isq) : R(0);
                R mt_e = R(0), dx_e = R(0);                                       // lane e: mt[e] (e < F), dX[e] (e >= F)
                if (isZ) { if (lane < F) mt_e = sg * vp; else if (lane < m) dx_e = sg * vp; }
                else {
                    if (lane < F) { const int r = lane + 1; mt_e = (r == kr) ? -sg : ((r == kr - 1) ? sg : R(0)); }      // -sg kvec
                    else if (lane > F && lane < m) { const int r = lane - F; dx_e = sg * ((r == kr) ? (kr >= 2 ? R(2) : R(1)) : ((r == kr - 1 || r == kr + 1) ? R(-1) : R(0))); }
                }
                if (lane < m) { L.vp[lane] = vp; L.mt[lane] = mt_e; }
                const R npn = isZ ? (dt * dt * (R)row + ((p_k1 >= 1 ? p_w1 * p_w1 : R(0)) + p_w2 * p_w2) * iQf) : (kr >= 2 ? R(2) : R(1));
                R mu_p = R(0);
                bool failed = false, fresh = true;                                // fresh: sviol still valid from the search
#ifndef ISMPC_A_FIRST_STEP
#define ISMPC_A_FIRST_STEP 1
#endif
                // ---- the FIRST row of a solve, in closed form (round 4).  At the equality-only point the working set holds the stability row
                // alone: no active ZMP row (no neighbours, no tridiagonal part, G is its constant part), no kinematic row.  The small system then
                // decouples -- the footstep unknowns are 0 and the stability unknown is cE = sg dt PA_row / a'a --, nothing can leave the working
                // set (t1 = infinity), and the whole step is z_u = sg dt [i <= row] - cE a, z_f = -sg M_row / Qf with the full step length
                // t = -violation / gamma, gamma = |n+|^2 - (dt PA_row)^2 / a'a.  These are the generic step's own numbers (same expressions) without
                // its machinery: ~250 vector instructions instead of ~950, on every one-shot solve and every cold restart.
                bool first_done = false;
                if (ISMPC_A_FIRST_STEP && isZ && qz == 0 && qk == 0) {
                    LANE_FRESH();
                    const R dE = sg * dt * p_pa;                                   // dX of the stability column
                    const R cE = dE * frcp((R)Dd);                                 // (Dd = a'a here: no active ZMP row)
                    const R gamma = npn - dE * cE;
                    if (gamma > (R)NM::gamma_rel * npn && iters < c.max_iter) {
                        ++iters;
                        WAVE_LDS_SYNC();                                           // L.vp is complete
                        const R t = -sviol / gamma;
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int i = lane * RL + k + 1;
                            if (i <= C) {
                                const R zu = -cE * ap[i - 1] + (i <= row ? sg * dt : R(0));
                                u[k] += t * zu;
                                if (i == row) { SET_STA_(k, sg > R(0) ? 1 : -1); mu[k] = t; }
                                if (i < row) SET_NXT_(k, row);                     // every earlier row sees `row` as its next active row,
                                if (i > row) SET_PRV_(k, row);                     // every later one as its previous
                            }
                        }
                        if (klane) fr += t * (-sg * L.vp[lane - 1] * sq) * iQf;
                        muE -= t * cE;
                        const R g1 = idt2 * rinv[row];
                        for (int e = lane; e < m * m; e += 64) { const int i = e / m, jj = e - i * m; L.G[e] += g1 * L.vp[i] * L.vp[jj]; }
                        WAVE_LDS_SYNC();
                        qz = 1;
                        Dd += gap_term(0, row) + tail_term(row) - tail_term(0);
                        first_done = true;
                    }
                }
#ifndef ISMPC_A_SECOND_STEP
#define ISMPC_A_SECOND_STEP 1
#endif
                // ---- the SECOND row, in closed form too (round 4; docs/models/proto_second_step.py checks these expressions against the general
                // structured solve).  Working set = stability row + ONE ZMP row j: G = g v v' with v = V_j = [M~_j, dt PA_j], g = 1 / (j dt^2), so
                // the small system is diag(I_F, -a'a) + g v v' and Sherman-Morrison gives its solution from a handful of wave-uniform scalars:
                //     kappa = sg w + g (M~_j . mt)            w = row / j (row < j) or 1 (row > j): interpolation weight of the new row at j
                //     beta = |M~_j|^2 - vE^2 / a'a ,  1 + g beta = D / a'a + g |M~_j|^2   (D = a'a - G_EE: the cancellation-free defect)
                //     alpha = kappa - g (kappa beta + vE dX / a'a) / (1 + g beta) ;  c_f = alpha M~_j ;  cE = (dX - alpha vE) / a'a
                //     rho_j = g (M~_j . mt - alpha beta - vE dX / a'a) + sg w
                // Taken only when the full step is the step (t2 <= t1: row j keeps its place); a partial step goes the generic way, untouched.
                if (ISMPC_A_SECOND_STEP && !first_done && isZ && qz == 1 && qk == 0 && iters < c.max_iter) {
                    LANE_FRESH();
                    const int pnr = at_row<int, RL>(pn, row);
                    const int na = PN_PRV(pnr), nb = PN_NXT(pnr);
                    const int jr = na > 0 ? na : nb;                               // the one active ZMP row
                    const int sj = ((at_row<int, RL>(pn, jr) >> 18) & 3) - 1;      // +1 lower bound, -1 upper bound
                    const R mu_j = at_row<R, RL>(mu, jr);
                    const int b1 = L.k1s[jr - 1]; const R w1j = L.w1s[jr - 1], w2j = R(1) - w1j;
                    const R vE = dt * pap[jr];
                    const R g = idt2 * rinv[jr];
                    R mm = R(0);                                                   // M_row . M_j over the footstep columns
                    { const int a1 = p_k1;
                      if (a1 >= 1) { if (a1 == b1) mm += p_w1 * w1j; else if (a1 == b1 + 1) mm += p_w1 * w2j; }
                      { const int cx = a1 + 1; if (cx == b1 && b1 >= 1) mm += p_w2 * w1j; else if (cx == b1 + 1) mm += p_w2 * w2j; } }
                    const R vmt = sg * mm * iQf;                                   // M~_j . mt
                    const R v11 = ((b1 >= 1 ? w1j * w1j : R(0)) + w2j * w2j) * iQf;   // |M~_j|^2
                    const R wint = (nb > 0) ? (R)row * rinv[nb] : R(1);
                    const R dX = sg * dt * p_pa;
                    const R iaa = frcp(aa);
                    const R kappa = sg * wint + g * vmt;
                    const R beta = v11 - vE * vE * iaa;
                    const R den = (R)Dd * iaa + g * v11;                           // 1 + g beta
                    const R xs = vE * dX * iaa;
                    const R alpha = kappa - g * (kappa * beta + xs) * frcp(den);
                    const R cE = (dX - alpha * vE) * iaa;
                    const R rho = g * (vmt - alpha * beta - xs) + sg * wint;       // unsigned: the multiplier of row j moves by -t sj rho
                    const R dj = sg * (dt * dt * (R)min(row, jr) + mm * iQf);
                    const R gamma = npn - (dj * rho + dX * cE);
                    const R rs = (sj > 0 ? R(1) : R(-1)) * rho;
                    const R t1 = (rs > R(0)) ? mu_j * frcp(rs) : R(INFINITY);
                    const R t2 = (gamma > (R)NM::gamma_rel * npn) ? -sviol / gamma : R(INFINITY);
                    if (den > R(0.05) && t2 < R(INFINITY) && t2 <= t1) {
                        ++iters;
                        const R t = t2;
                        const R vj_e = (lane < m) ? border_elem<R, F>(lane, b1, w1j, pap[jr], dt, isq) : R(0);     // V_j, one element per lane
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int i = lane * RL + k + 1;
                            if (i <= C) {
                                const R zu = (i <= row ? sg * dt : R(0)) - (i <= jr ? dt * rho : R(0)) - cE * ap[i - 1];
                                u[k] += t * zu;
                                if (i == jr) mu[k] -= t * rs;
                                if (i == row) { SET_STA_(k, sg > R(0) ? 1 : -1); mu[k] = t; }
                                if (i >= na && i < row) SET_NXT_(k, row);
                                if (i > row && (nb == 0 || i <= nb)) SET_PRV_(k, row);
                            }
                        }
                        if (klane) {
                            const R mrow = border_elem<R, F>(lane - 1, p_k1, p_w1, p_pa, dt, isq), mj = border_elem<R, F>(lane - 1, b1, w1j, pap[jr], dt, isq);
                            fr += t * (-sg * mrow * sq + sq * alpha * mj) * iQf;
                        }
                        muE -= t * cE;
                        const R va = na > 0 ? vj_e : R(0), vb = nb > 0 ? vj_e : R(0);
                        if (lane < m) { L.d1[lane] = vp - va; L.d2[lane] = vb - vp; L.d0[lane] = vb - va; }
                        WAVE_LDS_SYNC();
                        const R g1 = idt2 * rinv[row - na], g2 = nb > 0 ? idt2 * rinv[nb - row] : R(0), g0 = nb > 0 ? idt2 * rinv[nb - na] : R(0);
                        for (int e = lane; e < m * m; e += 64) {
                            const int i = e / m, jj = e - i * m;
                            L.G[e] += g1 * L.d1[i] * L.d1[jj] + g2 * L.d2[i] * L.d2[jj] - g0 * L.d0[i] * L.d0[jj];
                        }
                        WAVE_LDS_SYNC();
                        qz = 2;
                        Dd += (nb > 0) ? gap_term(na, row) + gap_term(row, nb) - gap_term(na, nb)
                                       : gap_term(na, row) + tail_term(row) - tail_term(na);
                        first_done = true;
                    }
                }
                // ================= steps until the row enters (Goldfarb-Idnani) =================
                if (!first_done)
                for (;;) {
                    LANE_FRESH();
                    if (++iters > c.max_iter) { status |= ISMPC_A_ST_ITER_LIMIT; failed = true; break; }
                    // ---- violation of the row at the current point (after a partial step)
                    if (!fresh) {
                        if (isZ) {
                            R vv[RL];
                            row_values(vv);
                            const R v = at_row<R, RL>(vv, row);
                            sviol = sg > R(0) ? v - zlo : zhi - v;
                        } else {
                            const R fprev = dppv<0x111, 0xf, true>(R(0), fr);
                            const R vk = sg > R(0) ? (fr - fprev) - klo : khi - (fr - fprev);
                            sviol = sq * rl(vk, kr);
                        }
                    }
                    fresh = false;
                    // ---- neighbours (na < row < nb) of a new ZMP row among the active ones; V there
                    int na = 0, nb = 0; R th = R(0), va = R(0), vb = R(0), vint = R(0);
                    if (isZ && qz > 0) {
                        const int pnr = at_row<int, RL>(pn, row);                  // prv / nxt are kept for every row, active or not
                        na = PN_PRV(pnr); nb = PN_NXT(pnr);
                        if (na > 0 && lane < m) va = border_elem<R, F>(lane, L.k1s[na - 1], L.w1s[na - 1], pap[na], dt, isq);
                        if (nb > 0 && lane < m) vb = border_elem<R, F>(lane, L.k1s[nb - 1], L.w1s[nb - 1], pap[nb], dt, isq);
                        if (nb == 0) { vint = va; th = R(0); }
                        else if (na == 0) { th = (R)row * rinv[nb]; vint = th * vb; }
                        else { th = (R)(row - na) * rinv[nb - na]; vint = va + th * (vb - va); }
                    }
                    // ---- small quasi-definite system  [[I+G11, G1x],[Gx1, Gxx - Sxx]] cc = [h1 ; hx - dX]
                    // unknown order: 0..F-1 footstep columns, F = stability row, F+1..2F = Khat_1..F (inactive: pinned to 0)
                    if (lane < m) {
                        R h_e = sg * vint;
#pragma unroll
                        for (int r = 0; r < F; ++r) h_e += L.G[lane * m + r] * L.mt[r];
                        if (lane < F) h_e -= L.mt[lane];                                // (L.G carries the identity of the footstep block: G alone is meant here)
                        L.hx[lane] = h_e - dx_e;
                    }
                    WAVE_LDS_SYNC();
                    const unsigned long long kmask = __builtin_amdgcn_ballot_w64(klane && kact != 0);   // bit r: Khat_r active
                    Dee = (R)Dd;
                    PH(12);                                // 12: Goldfarb-Idnani: one step (new row, neighbours, right-hand side ...)
                    const R cc_e = (kmask == 0ull) ? solve_small(std::true_type{}, 0ull) : solve_small(std::false_type{}, kmask);
                    const R cE = L.cc[F];
                    // ---- y = coefficients on the V columns (delta_Z - V cc = sg dt^2 k_i + V y); rows see the footstep
                    // columns through comb[k1], comb[k1+1]:  comb[r] = (yM_r - yK_r + yK_{r+1}) / sqrt(Qf)
                    if (lane <= F + 1) {
                        R cb = R(0);
                        if (klane) {
                            const int r = lane;
                            const R yM = L.mt[r - 1] - L.cc[r - 1], yK = -L.cc[F + r], yKn = (r + 1 <= F) ? -L.cc[F + r + 1] : R(0);
                            cb = (yM - yK + yKn) * isq;
                        }
                        L.comb[lane] = cb;
                    }
                    WAVE_LDS_SYNC();
                    R svl[RL];
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        const R w1s_ = W1_(k);
                        svl[k] = (i <= C) ? (w1s_ * L.comb[K1_(k)] + (R(1) - w1s_) * L.comb[K1_(k) + 1]) : R(0);   // without - dt PA_i cE (split off, as in block_solve)
                        if (i <= C) L.sv[i - 1] = svl[k];
                    }
                    WAVE_LDS_SYNC();
                    // ---- rho per active ZMP row (tridiagonal K^-1) + interpolation weights; d.r ; dual step length
                    R rho[RL], ddl = R(0), tcand = R(INFINITY); int tcode = 0;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        const int sk = STA_(k);
                        rho[k] = R(0);
                        if (i <= C && sk != 0) {
                            const int pv = PRV_(k), nx = NXT_(k);
                            const R sp = pv > 0 ? L.sv[pv - 1] : R(0);
                            R r_ = (svl[k] - sp) * rinv[i - pv] * idt2 - idt * cE * abar(pv, i);
                            if (nx > 0) r_ -= (L.sv[nx - 1] - svl[k]) * rinv[nx - i] * idt2 - idt * cE * abar(i, nx);
                            if (isZ) {
                                if (i == na) r_ += (nb == 0) ? sg : sg * (R(1) - th);
                                if (i == nb) r_ += sg * th;
                            }
                            rho[k] = r_;
                            const R w1k = W1_(k), w2k = R(1) - w1k;
                            const int b1 = K1_(k);
                            R dj;                                                  // sg <row+, Z_i>
                            if (isZ) {
                                R mm = R(0);                                       // M_p . M_i
                                const int a1 = p_k1;
                                if (a1 >= 1) { if (a1 == b1) mm += p_w1 * w1k; else if (a1 == b1 + 1) mm += p_w1 * w2k; }
                                { const int cx = a1 + 1; if (cx == b1 && b1 >= 1) mm += p_w2 * w1k; else if (cx == b1 + 1) mm += p_w2 * w2k; }
                                dj = sg * (dt * dt * (R)min(row, i) + mm * iQf);
                            } else {
                                R mk = R(0);                                       // M_i . kvec_kr
                                if (b1 == kr) mk += w1k;
                                if (b1 + 1 == kr) mk += w2k;
                                if (kr - 1 >= 1) { if (b1 == kr - 1) mk -= w1k; if (b1 + 1 == kr - 1) mk -= w2k; }
                                dj = sg * (-mk) * isq;
                            }
                            ddl += dj * r_;
                            const R rs = (sk > 0 ? R(1) : R(-1)) * r_;
                            if (rs > R(0)) { const R tt = mu[k] * frcp(rs); if (tt < tcand) { tcand = tt; tcode = i; } }
                        }
                    }
                    if (lane >= F && lane < m) ddl += dx_e * cc_e;                 // border part of d.r
                    const R cK = (klane) ? L.cc[F + lane] : R(0);                  // lane r: unsigned cc of Khat_r
                    if (klane && kact != 0) {
                        const R rs = (kact > 0 ? R(1) : R(-1)) * cK;
                        if (rs > R(0)) { const R tt = muK * frcp(rs); if (tt < tcand) { tcand = tt; tcode = C + lane; } }
                    }
                    const R gamma = npn - wave_sum(ddl);
                    const R t1 = wave_min(tcand);
                    int lrow = 0;
                    if (t1 < R(INFINITY)) lrow = rl(tcode, (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(tcand == t1)));
                    const R t2 = (gamma > (R)NM::gamma_rel * npn) ? -sviol / gamma : R(INFINITY);
                    const R t = fmin(t1, t2);
                    if (!(t < R(INFINITY))) { status |= (axis == 0 ? ISMPC_A_ST_X_INFEASIBLE : ISMPC_A_ST_Y_INFEASIBLE); failed = true; break; }
                    // ---- primal step: z_u = suffix sum of (-dt rho, + sg dt at the new row) - r_E a ; z_f from cc
                    if (t2 < R(INFINITY)) {
                        // z_u(i) = dt (sg [i <= row] - sum over active i' >= i of rho_i') - r_E a: the sum of the tridiagonal part
                        // telescopes to the slope of svl between the two active rows around i, the interpolation weights of the
                        // new row (1 - th at na, th at nb) and its own impulse are step functions -- no scan
                        const R wA = (isZ && na > 0) ? (nb == 0 ? R(1) : R(1) - th) : R(0), wB = (isZ && nb > 0) ? th : R(0);
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int i = lane * RL + k + 1;
                            if (i <= C) {
                                const int sk = STA_(k);
                                const int pv = PRV_(k), nx = NXT_(k);
                                const int jj = (sk != 0) ? i : nx;
                                R zu = -cE * ap[i - 1];
                                if (jj > 0) {
                                    const R sp = pv > 0 ? L.sv[pv - 1] : R(0);
                                    const R sj = (sk != 0) ? svl[k] : L.sv[jj - 1];
                                    zu = -(sj - sp) * rinv[jj - pv] * idt - cE * (ap[i - 1] - abar(pv, jj));   // the stability multiplier
                                }                                                                              // drops out of active runs
                                if (isZ) {
                                    const R stepf = (i <= row ? R(1) : R(0)) - (i <= na ? wA : R(0)) - (i <= nb ? wB : R(0));
                                    zu += sg * dt * stepf;
                                }
                                u[k] += t * zu;
                            }
                        }
                        if (klane) {
                            // z_f[r] = ( n+_f[r] + sqrt(Qf) c1[r] - sqrt(Qf) (cK[r] - cK[r+1]) ) / Qf
                            const int r = lane;
                            R nf = isZ ? -sg * L.vp[r - 1] * sq : ((r == kr) ? sg * sq : ((r == kr - 1) ? -sg * sq : R(0)));
                            nf += sq * L.cc[r - 1];
                            nf -= sq * cK;
                            if (r + 1 <= F) nf += sq * L.cc[F + r + 1];
                            fr += t * nf * iQf;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < RL; ++k) { const int sk = STA_(k); if (sk != 0) mu[k] -= t * (sk > 0 ? R(1) : R(-1)) * rho[k]; }
                    if (klane && kact != 0) muK -= t * (kact > 0 ? R(1) : R(-1)) * cK;
                    muE -= t * cE;
                    mu_p += t;
                    if (t2 < R(INFINITY) && t == t2) {
                        // ============ the row enters ============
                        if (isZ) {
                            if (lane < m) { L.d1[lane] = vp - va; L.d2[lane] = (nb > 0 ? vb : R(0)) - vp; L.d0[lane] = (nb > 0 ? vb : R(0)) - va; }
                            WAVE_LDS_SYNC();
                            const R g1 = idt2 * rinv[row - na], g2 = nb > 0 ? idt2 * rinv[nb - row] : R(0), g0 = nb > 0 ? idt2 * rinv[nb - na] : R(0);
                            for (int e = lane; e < m * m; e += 64) {
                                const int i = e / m, jj = e - i * m;
                                L.G[e] += g1 * L.d1[i] * L.d1[jj] + g2 * L.d2[i] * L.d2[jj] - g0 * L.d0[i] * L.d0[jj];
                            }
                            WAVE_LDS_SYNC();
#pragma unroll
                            for (int k = 0; k < RL; ++k) {
                                const int i = lane * RL + k + 1;
                                if (i == row) { SET_STA_(k, sg > R(0) ? 1 : -1); mu[k] = mu_p; }
                                if (i >= na && i < row) SET_NXT_(k, row);          // rows that now see `row` as their next / previous active row
                                if (i > row && (nb == 0 || i <= nb)) SET_PRV_(k, row);
                            }
                            ++qz;
                            Dd += (nb > 0) ? gap_term(na, row) + gap_term(row, nb) - gap_term(na, nb)
                                           : gap_term(na, row) + tail_term(row) - tail_term(na);
                        } else {
                            if (lane == kr) { kact = sg > R(0) ? 1 : -1; muK = mu_p; }
                            ++qk;
                        }
                        break;
                    }
                    // ============ partial step: working-set row lrow leaves ============
#ifdef ISMPC_A_DIAG
                    ++dg_part;
#endif
                    if (lrow <= C) {
                        const int pnl = at_row<int, RL>(pn, lrow);
                        const int pa_ = PN_PRV(pnl), pb_ = PN_NXT(pnl);
                        R vl_ = R(0), wa_ = R(0), wb_ = R(0);
                        if (lane < m) {
                            vl_ = border_elem<R, F>(lane, L.k1s[lrow - 1], L.w1s[lrow - 1], pap[lrow], dt, isq);
                            if (pa_ > 0) wa_ = border_elem<R, F>(lane, L.k1s[pa_ - 1], L.w1s[pa_ - 1], pap[pa_], dt, isq);
                            if (pb_ > 0) wb_ = border_elem<R, F>(lane, L.k1s[pb_ - 1], L.w1s[pb_ - 1], pap[pb_], dt, isq);
                            L.d1[lane] = vl_ - wa_; L.d2[lane] = (pb_ > 0 ? wb_ : R(0)) - vl_; L.d0[lane] = (pb_ > 0 ? wb_ : R(0)) - wa_;
                        }
                        WAVE_LDS_SYNC();
                        const R g1 = idt2 * rinv[lrow - pa_], g2 = pb_ > 0 ? idt2 * rinv[pb_ - lrow] : R(0), g0 = pb_ > 0 ? idt2 * rinv[pb_ - pa_] : R(0);
                        for (int e = lane; e < m * m; e += 64) {
                            const int i = e / m, jj = e - i * m;
                            L.G[e] -= g1 * L.d1[i] * L.d1[jj] + g2 * L.d2[i] * L.d2[jj] - g0 * L.d0[i] * L.d0[jj];
                        }
                        WAVE_LDS_SYNC();
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int i = lane * RL + k + 1;
                            if (i == lrow) { SET_STA_(k, 0); mu[k] = R(0); }
                            if (i >= pa_ && i < lrow) SET_NXT_(k, pb_);
                            if (i > lrow && (pb_ == 0 || i <= pb_)) SET_PRV_(k, pa_);
                        }
                        --qz;
                        Dd += (pb_ > 0) ? gap_term(pa_, pb_) - gap_term(pa_, lrow) - gap_term(lrow, pb_)
                                        : tail_term(pa_) - gap_term(pa_, lrow) - tail_term(lrow);
                    } else {
                        if (lane == lrow - C) { kact = 0; muK = R(0); }
                        --qk;
                    }
                }
                PH(12);                                    // ... (step lengths, primal / dual update, row enters / leaves)
                if (failed) { gi_leave = true; break; }
                ++gi_adds;
            }
            if (gi_leave || phase != 0) break;
            phase = 1;
            }
            // ---- every row, active or not, the kinematic rows and the stability row are checked once more at the point that
            // is about to be returned: a working set that pins (nearly) every variable can wear the incremental solves down
            // without any inactive row showing it.  One block solve of the final working set (kinematic rows included)
            // polishes such a point; if it still fails, the QP is reported infeasible (the reference's quadprog returns no
            // solution on infeasible QPs).
            PH(15);
            if (status == 0 && !done_opt && !defer_qp) {
                auto off_point = [&]() __attribute__((always_inline)) -> bool {
                    LANE_FRESH();
                    R vv[RL], aul = R(0);
                    row_values(vv);
                    bool bad = false;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int i = lane * RL + k + 1;
                        if (i <= C) {
                            const R v = vv[k];
                            const R tol = (R)NM::final_rel * (fabs(v) + fmax(fabs(zlo), fabs(zhi))) + (R)NM::final_abs;
                            bad = bad || !(v - zlo >= -tol && zhi - v >= -tol);
                            aul += ap[i - 1] * u[k];
                        }
                    }
                    const R fprev = dppv<0x111, 0xf, true>(R(0), fr);
                    if (klane && khi < R(INFINITY)) {
                        const R v = fr - fprev, tol = (R)NM::final_rel * (fabs(v) + fmax(fabs(klo), fabs(khi))) + (R)NM::final_abs;
                        bad = bad || !(v - klo >= -tol && khi - v >= -tol);
                    }
                    const R eqr = wave_sum(aul) - beq;
                    WAVE_LDS_SYNC();
                    return __builtin_amdgcn_ballot_w64(bad) != 0 || !(fabs(eqr) <= (R)NM::final_rel * (R(1) + fabs(beq)));
                };
                bool bad = off_point();
                if (bad && c.warm_add > 0) {
                    ++iters;
                    block_solve(0ull);                                           // kinematic rows stay out of the block phase (tried: the
                                                                                 // passes adding / dropping them too changes nothing on the bench workloads)
                    R mmax = fabs(muK);
#pragma unroll
                    for (int k = 0; k < RL; ++k) mmax = fmax(mmax, fabs(mu[k]));
                    const R mtol = (R)NM::mult_rel * (R(1) - wave_min(-mmax));
                    bool negm = klane && kact != 0 && muK < -mtol;
#pragma unroll
                    for (int k = 0; k < RL; ++k) negm = negm || (STA_(k) != 0 && mu[k] < -mtol);
                    bad = __builtin_amdgcn_ballot_w64(negm) != 0 || off_point();
                }
                if (bad) status |= (axis == 0 ? ISMPC_A_ST_X_INFEASIBLE : ISMPC_A_ST_Y_INFEASIBLE) | ISMPC_A_ST_UNVERIFIED;
            }
        }

        PH(13);                                            // 13: final check / polish
        if (defer_qp) {                                       // nothing of this QP is written here: the fp64 re-solve owns it
            if (lane == 0) defer_list[atomicAdd(defer_count, 1)] = qp;
            WAVE_LDS_SYNC();
            continue;
        }
        if (hist != nullptr) {
            unsigned long long* hq = hist + (size_t)qp * 8;
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const unsigned long long lo_ = __builtin_amdgcn_ballot_w64(status == 0 && STA_(k) > 0), hi_ = __builtin_amdgcn_ballot_w64(status == 0 && STA_(k) < 0);
                if (lane == 0) { hq[k] = lo_; hq[4 + k] = hi_; }
            }
        }
        LANE_FRESH();
        // ---- LIP update (:297-322), footstep bookkeeping (:522-556), outputs: fp64 whatever the precision of the solve
#ifdef ISMPC_A_DIAG
        // bits 0-9 work (as in the product build) | 10-13 block solves | 14-15 why the solve started cold (1 check, 2 budget) |
        // 16-23 ZMP rows active when Goldfarb-Idnani took over | 24-31 partial steps (rows that left inside Goldfarb-Idnani)
        iters = (iters & 1023) | ((dg_ns & 15) << 10) | ((dg_cold & 3) << 14) | ((dg_q0 & 255) << 16) | ((dg_part & 255) << 24);
#endif
        const bool ok = (status & (ISMPC_A_ST_BAD_INDEX | ISMPC_A_ST_OVERFLOW)) == 0;
        const double u0 = ok ? (double)rl(u[0], 0) : 0.0;
        const double df0 = ok ? (double)rl(fr, 1) : 0.0;                          // first footstep, relative to the current one
        if (lane == 0) {
            // The instance's record is READ AGAIN here, by this one lane, instead of staying live -- in scalar registers, it is wave-uniform --
            // across the whole solve: position, velocity, ZMP, current footstep, counters and (per-instance) the gait parameters were a
            // quarter of the scalar registers the solver phases had to spill around (round 4).  state_in is the launch's read-only copy.
            const ismpc_a_state* sp = state_in + inst;
            asm volatile("" : "+v"(sp));                                         // a per-lane address: a vector load of lane 0, not a scalar one
            const double p0 = axis == 0 ? sp->x : sp->y, z0 = axis == 0 ? sp->xz : sp->yz, cur2 = axis == 0 ? sp->cur_x : sp->cur_y;
            double v0 = axis == 0 ? sp->xd : sp->yd;
            if (push) { const double* pp = push + inst * 2 + axis; asm volatile("" : "+v"(pp)); v0 += *pp; }
            const int j2 = sp->j, fc2 = sp->fc;
            int step2 = c.step; double eta2 = c.eta;
            const PiPre* ppre = nullptr;
            if (PI) {
                const ismpc_a_inst* ipp = ipar + inst; asm volatile("" : "+v"(ipp));
                ppre = pre + inst; asm volatile("" : "+v"(ppre));
                if (!(status & ISMPC_A_ST_BAD_INDEX)) { step2 = ipp->step; eta2 = ppre->eta; }
            }
            const double f0 = cur2 + df0;
            double np_, nv_, nz_;
            if (PI) {                                                            // A_upd, B_upd for this instance's eta (:67-71)
                const double ch = ppre->ch, sh = ppre->sh, she = ppre->sh_eta;      // cosh, sinh of eta dt (tick prologue)
                np_ = (ch * p0 + she * v0 + (1 - ch) * z0) + (c.dt - she) * u0;
                nv_ = ((eta2 * sh) * p0 + ch * v0 + (-eta2 * sh) * z0) + (1 - ch) * u0;
                nz_ = (0.0 * p0 + 0.0 * v0 + 1.0 * z0) + c.dt * u0;
            } else {
                np_ = (c.Au[0] * p0 + c.Au[1] * v0 + c.Au[2] * z0) + c.Bu[0] * u0;
                nv_ = (c.Au[3] * p0 + c.Au[4] * v0 + c.Au[5] * z0) + c.Bu[1] * u0;
                nz_ = (c.Au[6] * p0 + c.Au[7] * v0 + c.Au[8] * z0) + c.Bu[2] * u0;
            }
            ismpc_a_state* so = state + inst;
            const bool stepped = ok && (j2 + 1 >= step2 * fc2);
            if (ok) {
                if (axis == 0) { so->x = np_; so->xd = nv_; so->xz = nz_; } else { so->y = np_; so->yd = nv_; so->yz = nz_; }
                if (stepped) {
                    const double noff = f0 - fs[fc2];
                    if (axis == 0) { so->cur_x = f0; so->off_x = noff; } else { so->cur_y = f0; so->off_y = noff; }
                }
                if (axis == 0) { so->j = j2 + 1; if (stepped) { so->fc = fc2 + 1; so->rebuilt = 1; } }
            }
            if (out) {
                ismpc_a_out* o = out + inst;
                const int q = 1 + qz + qk;
                o->com_before[axis] = p0; o->vel_after[axis] = ok ? nv_ : v0; o->u0[axis] = u0; o->f0[axis] = f0;
                if (axis == 0) { o->iters_x = iters; atomicOr(&o->status, status); atomicOr(&o->active, q & 0xffff); }
                else { o->iters_y = iters; atomicOr(&o->status, status); atomicOr(&o->active, (q & 0xffff) << 16); }
            }
        }
        WAVE_LDS_SYNC();
        PH(14);                                            // 14: history, LIP update, outputs
        PH_FLUSH();
#undef W1_
#undef PN_RESET_
#undef PN_PRV
#undef PN_NXT
#undef K1_
#undef STA_
#undef SET_STA_
#undef PRV_
#undef NXT_
#undef SET_PRV_
#undef SET_NXT_
    }
}

// ---- launch: persistent grid = exactly the workgroups that are resident at once (registers / LDS decide how many per CU)
template <typename R, int RL, int F, bool PI>
inline int launch_one(const WaveLaunch& L, hipError_t* err)
{
    auto kern = ismpc_a_tick_wave<R, RL, F, PI>;
    int& occ = L.occ_cache[(F - 3) * 4 + (sizeof(R) == 4 ? 2 : 0) + (PI ? 1 : 0)];
    if (occ == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, WG, 0) != hipSuccess || nb < 1) nb = 1;
        occ = nb;
    }
    int grid = std::min((2 * L.batch + 3) / 4, L.cus * occ);
    if (L.grid_cap > 0) grid = std::min(grid, L.grid_cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WG), 0, L.stream, L.c_dev, L.prev, L.state, L.inst, L.push, L.out, L.batch, L.work_counter, L.hist, L.hist_load,
                       L.order, L.count_ptr, L.claim_chunk, L.static_q, L.order_is_qp, L.defer_list, L.defer_count, L.pre);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = e; return -2; }
    return 0;
}
template <typename R, int RL, int F>
inline int launch_pi(const WaveLaunch& L, hipError_t* err) { return L.inst ? launch_one<R, RL, F, true>(L, err) : launch_one<R, RL, F, false>(L, err); }
template <typename R, int RL>
inline int launch_f(const WaveLaunch& L, hipError_t* err)
{
    switch (L.F) {
        case 3: return launch_pi<R, RL, 3>(L, err);
        case 4: return launch_pi<R, RL, 4>(L, err);
        case 5: return launch_pi<R, RL, 5>(L, err);
        case 6: return launch_pi<R, RL, 6>(L, err);
        default: return -1;
    }
}
template <int RL>
inline int launch_wave(const WaveLaunch& L, hipError_t* err) { return L.precision == 1 ? launch_f<float, RL>(L, err) : launch_f<double, RL>(L, err); }

}  // namespace ismpc_a
