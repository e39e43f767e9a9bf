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
#include <string>
#include <vector>
#include <new>
#include "ismpc_tables.hpp"

namespace {

constexpr int TI = 16;          // instances per workgroup = MFMA M tile
constexpr int WAVES = 4;        // wavefronts per workgroup
constexpr int IPW = TI / WAVES; // instances each wavefront walks through in phases A and C

typedef double d4 __attribute__((ext_vector_type(4)));

struct DevConst {
    int N, NP, NPs, S, F, nmid, npat, Fmax, rows, tick_divisor;
    double dt, cdt, mass, g, h_des, half_run, half_first, q_p, q_u, q_v, z_lo, z_hi, gate, eta;
    const double *Hinv, *W, *midx, *midy, *midz, *tailx, *taily, *ftsp_t;
    const int *e_lo, *ne;
};

// ---- wavefront (64 lanes) primitives -------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over lanes strictly above this one
__device__ __forceinline__ double wave_suffix_excl(double v, int lane)
{
    double s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { double t = __shfl_down(s, o); if (lane + o < 64) s += t; }
    return s - v;
}
// sum over lanes strictly below this one
__device__ __forceinline__ double wave_prefix_excl(double v, int lane)
{
    double s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { double t = __shfl_up(s, o); if (lane >= o) s += t; }
    return s - v;
}
__device__ __forceinline__ double bcast0(double v) { return __shfl(v, 0); }

struct M2 { double a, b, c, d; };   // [a b; c d]
__device__ __forceinline__ M2 mul(const M2& x, const M2& y)
{
    M2 r;
    r.a = x.a*y.a + x.b*y.c; r.b = x.a*y.b + x.b*y.d;
    r.c = x.c*y.a + x.d*y.c; r.d = x.c*y.b + x.d*y.d;
    return r;
}

// Caller bookkeeping in front of solve(): Controller.cpp:297-304 (enabled) and :310.
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
    const double t = w.sim / (c.dt / c.cdt);
    if (!(t > -1.0) || !(t < 2.0e9)) return ISMPC_ST_BAD_INDEX;
    idx = (int)t;
    if (idx < 0 || idx + 2 * c.N > c.nmid || w.mpc < 0) return ISMPC_ST_BAD_INDEX;
    return 0;
}

template <int R>
__global__ __launch_bounds__(256)
void ismpc_tick_kernel(const DevConst c, const ismpc_tick_in* __restrict__ in_ro, ismpc_tick_in* state_rw,
                       ismpc_tick_out* __restrict__ out, double* __restrict__ u_traj, int batch, int rollout_frame)
{
    extern __shared__ double smem[];                  // [TI][NPs]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int inst0 = blockIdx.x * TI;
    const int N = c.N, NP = c.NP, NPs = c.NPs;
    const double dt = c.dt;
    const ismpc_tick_in* in = (rollout_frame >= 0) ? state_rw : in_ro;

    // ---------------- phase A: f_z, MPCSolver.cpp:259 ----------------
    for (int q = 0; q < IPW; ++q) {
        const int li = wave * IPW + q;
        const int gi = inst0 + li;
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
                    const int n = lane * R + r;
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
                const double up = wave_suffix_excl(lp, lane);
                const double uv = wave_suffix_excl(lv, lane);
                double vt[R], lt = 0.0;
#pragma unroll
                for (int r = R - 1; r >= 0; --r) { tp[r] += up; vt[r] = lt; lt += tp[r]; }
                const double ut = wave_suffix_excl(lt, lane);
                const double cs = dt * dt / c.mass, cv = dt / c.mass;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int n = lane * R + r;
                    if (n < N) f[r] = c.q_p * cs * (vt[r] + ut) + c.q_v * cv * (tv[r] + uv) - c.q_u * c.mass * c.g;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { const int n = lane * R + r; if (n < NP) smem[li * NPs + n] = f[r]; }
    }
    __syncthreads();

    // ---------------- phase B: U = -F Hinv on the matrix cores ----------------
    {
        constexpr int MAXT = 4;                        // NP <= 256 -> 16 column tiles / 4 waves
        const int ntiles = NP >> 4;
        d4 acc[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        const int arow = lane & 15, kq = lane >> 4;
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
        double ux_tr[R], uy_tr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { ux_tr[r] = 0.0; uy_tr[r] = 0.0; }

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
            const double pc = wave_prefix_excl(lc, lane);
            double di[R], ld_ = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) { ci[r] += pc; di[r] = ld_; ld_ += ci[r]; }
            const double pd = wave_prefix_excl(ld_, lane);
            const double cs = dt * dt / c.mass;
            bool viol = false;
            double lam[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = lane * R + r;
                const double k = (double)n;
                const double su = cs * (di[r] + pd);
                if (n < N && (su < c.z_lo || su > c.z_hi)) viol = true;
                const double zpos = su + (z0 + (k + 1.0) * dt * zd0) - c.g * dt * dt * (0.5 * k * (k + 1.0));
                const double zacc = (1.0 / c.mass) * u[r] - c.g;
                lam[r] = (c.g + zacc) / zpos;                               // MPCSolver.cpp:306
            }
            if (__any(viol)) status |= ISMPC_ST_Z_INEQ_ACTIVE;
            uz0 = bcast0(u[0]);
            // ---- z integration, MPCSolver.cpp:274-278
            o_z = z0 + dt * zd0;
            o_zd = zd0 + (dt / c.mass) * uz0 - dt * c.g;
            if (isnan(o_z)) { o_z = c.h_des; status |= ISMPC_ST_Z_NAN; }
            if (isnan(o_zd)) { o_zd = 0.0; status |= ISMPC_ST_Z_NAN; }

            // ---- A_j, B_j per sample, MPCSolver.cpp:353-361
            M2 A[R]; double B0[R], B1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = lane * R + r;
                if (n >= N) { A[r] = (M2){1.0, 0.0, 0.0, 1.0}; B0[r] = 0.0; B1[r] = 0.0; }
                else if (lam[r] < c.gate) { A[r] = (M2){1.0, dt, 0.0, 1.0}; B0[r] = 0.0; B1[r] = 0.0; }
                else {
                    const double sq = sqrt(lam[r]);
                    const double ch = cosh(sq * dt), sh = sinh(sq * dt);
                    A[r] = (M2){ch, sh / sq, sq * sh, ch};
                    B0[r] = 1.0 - ch; B1[r] = -sq * sh;
                }
            }
            const double lam0 = bcast0(lam[0]);
            const M2 A0 = (M2){bcast0(A[0].a), bcast0(A[0].b), bcast0(A[0].c), bcast0(A[0].d)};
            const double B00 = bcast0(B0[0]), B10 = bcast0(B1[0]);

            if (lam0 > c.gate) {                                           // MPCSolver.cpp:322
                // ---- suffix products: X_lane = A_{N-1} ... A_{first sample of lane+1}
                M2 Y = A[0];
#pragma unroll
                for (int r = 1; r < R; ++r) Y = mul(A[r], Y);
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    M2 T; T.a = __shfl_down(Y.a, o); T.b = __shfl_down(Y.b, o); T.c = __shfl_down(Y.c, o); T.d = __shfl_down(Y.d, o);
                    if (lane + o < 64) Y = mul(T, Y);
                }
                M2 X; X.a = __shfl_down(Y.a, 1); X.b = __shfl_down(Y.b, 1); X.c = __shfl_down(Y.c, 1); X.d = __shfl_down(Y.d, 1);
                if (lane == 63) X = (M2){1.0, 0.0, 0.0, 1.0};
                // row vector c_n = C_sc A_{N-1} ... A_{n+1},  C_sc = [1, 1/eta]  (MPCSolver.cpp:375-379)
                const double ie = 1.0 / c.eta;
                double c0 = X.a + ie * X.c, c1 = X.b + ie * X.d;
                double a[R];
#pragma unroll
                for (int r = R - 1; r >= 0; --r) {
                    a[r] = c0 * B0[r] + c1 * B1[r];                        // Aeq(n) = C_sc phi_input(:,n)
                    const double n0 = c0 * A[r].a + c1 * A[r].c, n1 = c0 * A[r].b + c1 * A[r].d;
                    c0 = n0; c1 = n1;
                }
                const double cps0 = bcast0(c0), cps1 = bcast0(c1);           // C_sc phi_state
                // ---- box midpoints and reductions
                const double h = (w.fc > 1) ? c.half_run : c.half_first;     // MPCSolver.cpp:328-338
                double mx[R], my[R], aa[R];
                double s_abs = 0.0, s_sq = 0.0, s_ax = 0.0, s_ay = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int n = lane * R + r;
                    if (n < N) { mx[r] = c.midx[idx + n]; my[r] = c.midy[idx + n]; } else { mx[r] = 0.0; my[r] = 0.0; a[r] = 0.0; }
                    aa[r] = fabs(a[r]);
                    s_abs += aa[r]; s_sq += a[r] * a[r]; s_ax += a[r] * mx[r]; s_ay += a[r] * my[r];
                }
                s_abs = wave_sum(s_abs); s_sq = wave_sum(s_sq); s_ax = wave_sum(s_ax); s_ay = wave_sum(s_ay);
                const double beq_x = -(cps0 * x0 + cps1 * xd0) + c.tailx[idx];   // MPCSolver.cpp:381-384
                const double beq_y = -(cps0 * y0 + cps1 * yd0) + c.taily[idx];
                // v = u - mid:  sum a v = bp,  |v| <= h   ->  v_n = sg * sign(a_n) * min(tau |a_n|, h)
                const double bp[2] = { beq_x - s_ax, beq_y - s_ay };
                const double gmax = h * s_abs;
                double tau[2]; int its[2]; bool infeas[2];
#pragma unroll
                for (int ax = 0; ax < 2; ++ax) {
                    const double T = fabs(bp[ax]);
                    infeas[ax] = T > gmax * (1.0 + 1e-12) + 1e-300;
                    double t = 0.0; int prev = -1, it = 0;
                    if (infeas[ax]) t = INFINITY;
                    else {
                        for (; it < N + 2; ++it) {
                            double ssat = 0.0, qfree = 0.0; int cnt = 0;
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const bool sat = t * aa[r] >= h;
                                ssat += sat ? aa[r] : 0.0;
                                qfree += sat ? 0.0 : a[r] * a[r];
                                cnt += __popcll(__ballot(sat));
                            }
                            if (cnt == prev) break;
                            ssat = wave_sum(ssat); qfree = wave_sum(qfree);
                            if (!(qfree > 0.0)) { t = INFINITY; break; }
                            const double tn = (T - h * ssat) / qfree;
                            if (!(tn > t)) break;
                            t = tn; prev = cnt;
                        }
                    }
                    tau[ax] = t; its[ax] = it;
                }
                itx = its[0]; ity = its[1];
                if (infeas[0]) status |= ISMPC_ST_X_INFEASIBLE;
                if (infeas[1]) status |= ISMPC_ST_Y_INFEASIBLE;
                const double sgx = (bp[0] < 0.0) ? -1.0 : 1.0, sgy = (bp[1] < 0.0) ? -1.0 : 1.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double sa = (a[r] < 0.0) ? -1.0 : 1.0;
                    const double vx = (aa[r] > 0.0) ? fmin(tau[0] * aa[r], h) : 0.0;
                    const double vy = (aa[r] > 0.0) ? fmin(tau[1] * aa[r], h) : 0.0;
                    ux_tr[r] = mx[r] + sgx * sa * vx;
                    uy_tr[r] = my[r] + sgy * sa * vy;
                }
                ux0 = bcast0(ux_tr[0]); uy0 = bcast0(uy_tr[0]);
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
                if (n < N) { dst[n] = u[r]; dst[N + n] = ux_tr[r]; dst[2 * N + n] = uy_tr[r]; }
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

// ------------------------------------------------------------------------
thread_local std::string g_err = "";
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(ISMPC_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

}  // namespace

struct ismpc_handle {
    ismpc::Tables t;
    DevConst c{};
    int device = 0;
    std::vector<void*> dev_allocs;
    // staging for the host-pointer entry point
    ismpc_tick_in* st_in = nullptr; ismpc_tick_out* st_out = nullptr; int st_cap = 0;
    hipStream_t own_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false; bool timed_pending = false; double last_ms = 0.0;
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

int launch(ismpc_handle* h, int batch, const ismpc_tick_in* in, ismpc_tick_in* state, ismpc_tick_out* out,
           double* u_traj, int rollout_frame, hipStream_t s)
{
    if (batch <= 0) return ISMPC_OK;
    const int R = (h->c.N + 63) / 64;
    const dim3 grid((batch + TI - 1) / TI), block(64 * WAVES);
    const size_t lds = (size_t)TI * h->c.NPs * sizeof(double);
    switch (R) {
        case 1: hipLaunchKernelGGL(ismpc_tick_kernel<1>, grid, block, lds, s, h->c, in, state, out, u_traj, batch, rollout_frame); break;
        case 2: hipLaunchKernelGGL(ismpc_tick_kernel<2>, grid, block, lds, s, h->c, in, state, out, u_traj, batch, rollout_frame); break;
        case 3: hipLaunchKernelGGL(ismpc_tick_kernel<3>, grid, block, lds, s, h->c, in, state, out, u_traj, batch, rollout_frame); break;
        case 4: hipLaunchKernelGGL(ismpc_tick_kernel<4>, grid, block, lds, s, h->c, in, state, out, u_traj, batch, rollout_frame); break;
        default: return fail(ISMPC_E_UNSUPPORTED, "horizon N > 256");
    }
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

int ismpc_create(const ismpc_params* params, const double* ftsp, int rows, int device, ismpc_handle** out)
{
    if (!params || !ftsp || !out) return fail(ISMPC_E_INVALID, "null argument");
    *out = nullptr;
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
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete h; return fail(ISMPC_E_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e)); }
    const ismpc::Tables& t = h->t;
    DevConst& c = h->c;
    c.N = t.p.N; c.NP = t.NP; c.NPs = t.NP + 2; c.S = t.p.S; c.F = t.p.F; c.nmid = t.nmid; c.npat = t.npat;
    c.Fmax = t.Fmax; c.rows = t.rows; c.tick_divisor = t.tick_divisor;
    c.dt = t.p.mpc_dt; c.cdt = t.p.control_dt; c.mass = t.p.mass; c.g = t.p.g; c.h_des = t.p.h_des;
    c.half_run = t.p.foot_width / 2; c.half_first = t.p.first_step_halfwidth;
    c.q_p = t.p.q_p; c.q_u = t.p.q_u; c.q_v = t.p.q_v; c.z_lo = t.p.z_ineq_lo; c.z_hi = t.p.z_ineq_hi;
    c.gate = t.p.lambda_gate; c.eta = t.eta;
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
    if (rc != ISMPC_OK) { ismpc_destroy(h); return rc; }
    if (hipStreamCreate(&h->own_stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
        hipEventCreate(&h->ev1) != hipSuccess) { ismpc_destroy(h); return fail(ISMPC_E_NO_DEVICE, "stream/event creation failed"); }
    *out = h;
    return ISMPC_OK;
}

void ismpc_destroy(ismpc_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (void* p : h->dev_allocs) (void)hipFree(p);
    if (h->st_in) (void)hipFree(h->st_in);
    if (h->st_out) (void)hipFree(h->st_out);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int ismpc_solve_batch_device(ismpc_handle* h, int batch, const ismpc_tick_in* in_dev, ismpc_tick_out* out_dev,
                             double* u_traj, void* stream)
{
    if (!h || batch < 0 || (batch > 0 && (!in_dev || !out_dev))) return fail(ISMPC_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
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
    HIP_TRY(hipSetDevice(h->device));
    if (batch > h->st_cap) {
        if (h->st_in) (void)hipFree(h->st_in);
        if (h->st_out) (void)hipFree(h->st_out);
        h->st_in = nullptr; h->st_out = nullptr; h->st_cap = 0;
        HIP_TRY(hipMalloc((void**)&h->st_in, sizeof(ismpc_tick_in) * (size_t)batch));
        HIP_TRY(hipMalloc((void**)&h->st_out, sizeof(ismpc_tick_out) * (size_t)batch));
        h->st_cap = batch;
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
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->timing) HIP_TRY(hipEventRecord(h->ev0, s));
    for (int t = 0; t < ticks; ++t) {
        int rc = launch(h, batch, nullptr, state_dev, traj_dev ? traj_dev + (size_t)t * batch : nullptr, nullptr, first_frame + t, s);
        if (rc != ISMPC_OK) return rc;
    }
    if (h->timing) { HIP_TRY(hipEventRecord(h->ev1, s)); h->timed_pending = true; }
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
