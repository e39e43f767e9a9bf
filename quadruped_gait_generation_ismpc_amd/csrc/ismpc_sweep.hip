// Device-side table build for parameter sweeps of Formulation B (see ismpc_sweep.hpp).  What is computed follows
// MPCSolver::MPCSolver / the constant part of MPCSolver::solve (reference AMR_code_DART/MPCSolver.cpp:144-160,223-259) exactly
// as csrc/ismpc_tables.cpp restates it for one parameter set on the host; HOW differs: all sets at once, the N^3 work as
// batched MFMA products.
#include "ismpc_sweep.hpp"
#include <cmath>
#include <algorithm>

namespace ismpc {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NT = Tables::NT;
constexpr int PAR = 8;                  // doubles per set in SweepSlabs::par

// ---- H_k (MPCSolver.cpp:258 from the closed forms of :144-154), identity on the padding, and X0 = I / bound
__global__ __launch_bounds__(256) void sweep_init(const double* __restrict__ par, double* __restrict__ H, double* __restrict__ X, size_t smat,
                                                  int N, int NG, double dt)
{
    const int set = blockIdx.x;
    const double mass = par[set * PAR + 0], q_p = par[set * PAR + 1], q_u = par[set * PAR + 2], q_v = par[set * PAR + 3], scale = par[set * PAR + 6];
    const double cs = dt * dt / mass, cv = dt / mass;
    double* Hs = H + (size_t)set * smat; double* Xs = X + (size_t)set * smat;
    for (int e = threadIdx.x; e < NG * NG; e += blockDim.x) {
        const int i = e / NG, j = e - i * NG;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < N && j < N) {
            const int M = max(i, j);
            double a = 0.0;
            for (int k = M + 1; k < N; ++k) a += (double)(k - i) * (double)(k - j);       // S_bar_z' S_bar_z / cs^2 (exact integers)
            const double b = (double)(N - 1 - M);                                        // S_bar_z_v' S_bar_z_v / cv^2
            v = q_p * cs * cs * a + q_v * cv * cv * b + ((i == j) ? q_u : 0.0);
        }
        Hs[e] = v;
        Xs[e] = (i == j) ? scale : 0.0;
    }
}

// U = m S_bar_z' and Ut = m S_bar_z (unit mass; common to the sets)
__global__ __launch_bounds__(256) void sweep_unit_s(double* __restrict__ U, double* __restrict__ Ut, int N, int NG, double dt)
{
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < NG * NG; e += gridDim.x * blockDim.x) {
        const int j = e / NG, k = e - j * NG;
        U[e] = (j < k && k < N) ? (double)(k - j) * dt * dt : 0.0;          // U[j][k] = m S[k][j]
        Ut[e] = (k < j && j < N) ? (double)(j - k) * dt * dt : 0.0;         // Ut[j][k] = m S[j][k]
    }
}

// ---- batched C_k = op(alpha_k A_k B_k), 64 x 64 output tile per workgroup, four wavefronts of 2 x 2 MFMA tiles each,
// operands staged through LDS as 64 x 16 / 16 x 64 panels.  MODE 0: C = 2I - A B (Newton residual), 1: C = alpha A B,
// 2: C' = alpha A B stored TRANSPOSED with leading dimension NT, rows / columns < N only (the fallback tables' layout).
// v_mfma_f64_16x16x4_f64: A operand lane l = A[l & 15][l >> 4], B operand lane l = B[l >> 4][l & 15], D register v of lane l =
// D[(l >> 4) + 4 v][l & 15].
template <int MODE>
__global__ __launch_bounds__(256) void sweep_gemm(const double* __restrict__ A, size_t sA, const double* __restrict__ B, size_t sB,
                                                  double* __restrict__ C, size_t sC, const double* __restrict__ par, int alpha_inv_mass, int NG, int N)
{
    __shared__ double As[64][17];
    __shared__ double Bs[16][65];
    const int set = blockIdx.z, br = blockIdx.y * 64, bc = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const double* Ap = A + (size_t)set * sA; const double* Bp = B + (size_t)set * sB;
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    const int ar = tid >> 2, ac = (tid & 3) * 4;            // A panel: 64 rows x 16 k, four consecutive k per thread
    const int bk = tid >> 4, bcol = (tid & 15) * 4;         // B panel: 16 k x 64 columns, four consecutive columns per thread
    // software pipeline: the next panels travel from L2 to registers while the matrix cores work on the ones in LDS (one wavefront
    // per SIMD at these grid sizes: nothing else would hide the load latency)
    double ra[4], rb[4];
    {
        const double* ag = Ap + (size_t)(br + ar) * NG + ac;
        const double* bg = Bp + (size_t)bk * NG + bc + bcol;
#pragma unroll
        for (int t = 0; t < 4; ++t) { ra[t] = ag[t]; rb[t] = bg[t]; }
    }
    for (int kb = 0; kb < NG; kb += 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { As[ar][ac + t] = ra[t]; Bs[bk][bcol + t] = rb[t]; }
        __syncthreads();
        if (kb + 16 < NG) {
            const double* ag = Ap + (size_t)(br + ar) * NG + kb + 16 + ac;
            const double* bg = Bp + (size_t)(kb + 16 + bk) * NG + bc + bcol;
#pragma unroll
            for (int t = 0; t < 4; ++t) { ra[t] = ag[t]; rb[t] = bg[t]; }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kq = s * 4 + (lane >> 4);
            const double a0 = As[wr * 32 + (lane & 15)][kq], a1 = As[wr * 32 + 16 + (lane & 15)][kq];
            const double b0 = Bs[kq][wc * 32 + (lane & 15)], b1 = Bs[kq][wc * 32 + 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    const double alpha = alpha_inv_mass ? 1.0 / par[set * PAR + 0] : 1.0;
    double* Cp = C + (size_t)set * sC;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int gr = br + wr * 32 + i * 16 + (lane >> 4) + 4 * v, gc = bc + wc * 32 + j * 16 + (lane & 15);
                double val = alpha * acc[i][j][v];
                if (MODE == 0) val = ((gr == gc) ? 2.0 : 0.0) - val;
                if (MODE == 2) { if (gr < N && gc < N) Cp[(size_t)gc * NT + gr] = val; }
                else Cp[(size_t)gr * NG + gc] = val;
            }
}

// R = alpha_k U - T (the residual of H M1 = alpha U, T = H M1 from the MFMA product) and M1 += D (the correction Hinv R)
__global__ __launch_bounds__(256) void sweep_residual(const double* __restrict__ par, const double* __restrict__ U, double* __restrict__ T, size_t smat, int NG)
{
    const double alpha = 1.0 / par[blockIdx.y * PAR + 0];
    double* Ts = T + (size_t)blockIdx.y * smat;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < NG * NG; e += gridDim.x * blockDim.x) Ts[e] = fma(alpha, U[e], -Ts[e]);
}
__global__ __launch_bounds__(256) void sweep_accumulate(double* __restrict__ M, const double* __restrict__ D, size_t smat, int NG)
{
    double* Ms = M + (size_t)blockIdx.y * smat; const double* Ds = D + (size_t)blockIdx.y * smat;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < NG * NG; e += gridDim.x * blockDim.x) Ms[e] += Ds[e];
}

// X <- (X + X') / 2 on the N x N block (the iteration keeps X symmetric to rounding only)
__global__ __launch_bounds__(256) void sweep_symmetrise(double* __restrict__ X, size_t smat, int N, int NG)
{
    double* Xs = X + (size_t)blockIdx.y * smat;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N * N; e += gridDim.x * blockDim.x) {
        const int i = e / N, j = e - i * N;
        if (i < j) { const double v = 0.5 * (Xs[(size_t)i * NG + j] + Xs[(size_t)j * NG + i]); Xs[(size_t)i * NG + j] = v; Xs[(size_t)j * NG + i] = v; }
    }
}

// HSt[k][n] = M1[n][k]
__global__ __launch_bounds__(256) void sweep_transpose_nt(const double* __restrict__ M, size_t sM, double* __restrict__ out, size_t sO, int N, int NG)
{
    const double* Ms = M + (size_t)blockIdx.y * sM; double* Os = out + (size_t)blockIdx.y * sO;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N * N; e += gridDim.x * blockDim.x) {
        const int k = e / N, n = e - k * N;
        Os[(size_t)k * NT + n] = Ms[(size_t)n * NG + k];
    }
}

// ---- h_b = Hinv f_b, b = 0, a, b: f = f0 + z fa + zdot fb (MPCSolver.cpp:259; ismpc_tables.cpp "affine tables")
__global__ __launch_bounds__(256) void sweep_hvec(const double* __restrict__ par, const double* __restrict__ Hinv, const double* __restrict__ H, size_t smat,
                                                  double* __restrict__ hvec, int N, int NG, double dt)
{
    __shared__ double f[3][256], hcur[3][256], res[3][256];
    const int set = blockIdx.x, tid = threadIdx.x;
    const double mass = par[set * PAR + 0], q_p = par[set * PAR + 1], q_u = par[set * PAR + 2], q_v = par[set * PAR + 3], h_des = par[set * PAR + 4], g = par[set * PAR + 7];
    const double cs = dt * dt / mass, cv = dt / mass;
    if (tid < 3) {
        // S_bar_z' r (o[i] = cs sum_{k>i} (k-i) r_k) and S_bar_z_v' r (o[i] = cv sum_{k>i} r_k) as suffix sums
        double t1 = 0.0, t2 = 0.0, acc = 0.0;
        for (int i = N - 1; i >= 0; --i) {
            const double k = (double)i;
            const double tgz = -g * dt * dt * (k * (k + 1.0) / 2.0), tgv = -g * dt * k, tvec = (k + 1.0) * dt;
            double rp, rv;                                                   // position row / velocity row right-hand sides
            if (tid == 0) { rp = tgz - h_des; rv = tgv; } else if (tid == 1) { rp = 1.0; rv = 0.0; } else { rp = tvec; rv = 1.0; }
            double v = q_p * (cs * t2) + q_v * (cv * acc);
            if (tid == 0) v -= q_u * mass * g;
            f[tid][i] = v;
            t1 += rp; t2 += t1; acc += rv;
        }
    }
    __syncthreads();
    const double* Xs = Hinv + (size_t)set * smat; const double* Hm = H + (size_t)set * smat;
    for (int e = tid; e < 3 * N; e += blockDim.x) {
        const int b = e / N, i = e - b * N;
        double s = 0.0;
        for (int j = 0; j < N; ++j) s = fma(Xs[(size_t)i * NG + j], f[b][j], s);
        hcur[b][i] = s;
    }
    __syncthreads();
    // two steps of iterative refinement on H h = f: h = X f carries the inverse's own error, amplified by the cancellation in
    // X f (|X||f| >> |X f|); the corrected h only carries the rounding of the residual
    for (int step = 0; step < 2; ++step) {
        for (int e = tid; e < 3 * N; e += blockDim.x) {
            const int b = e / N, i = e - b * N;
            double r = f[b][i];
            for (int j = 0; j < N; ++j) r = fma(-Hm[(size_t)i * NG + j], hcur[b][j], r);
            res[b][i] = r;
        }
        __syncthreads();
        double upd[3] = {0.0, 0.0, 0.0};
        if (tid < N)
            for (int b = 0; b < 3; ++b) { double s = 0.0; for (int j = 0; j < N; ++j) s = fma(Xs[(size_t)tid * NG + j], res[b][j], s); upd[b] = s; }
        __syncthreads();
        if (tid < N) for (int b = 0; b < 3; ++b) hcur[b][tid] += upd[b];
        __syncthreads();
    }
    for (int e = tid; e < 3 * N; e += blockDim.x) hvec[((size_t)set * 3 + e / N) * NG + e % N] = hcur[e / N][e % N];
}

// o[k] = cs sum_{j<k} (k-j) v_j (S_bar_z applied as a double running sum), with the two running sums carried as double-double
// (TwoSum): the plain recurrence loses digits in sums of N^2/2 terms of mixed sign that the host's long double keeps
struct DD { double hi, lo; };
__device__ __forceinline__ void dd_add(DD& a, double b)
{
    const double s = a.hi + b, bb = s - a.hi, e = (a.hi - (s - bb)) + (b - bb);
    a.lo += e; const double t = s + a.lo; a.lo -= (t - s); a.hi = t;
}
__device__ __forceinline__ void dd_add(DD& a, const DD& b) { dd_add(a, b.hi); a.lo += b.lo; const double t = a.hi + a.lo; a.lo -= (t - a.hi); a.hi = t; }

// ---- one workgroup per (equality pattern, set): W_p = Hinv[:,E] (Hinv[E,E])^-1 (MPCSolver.cpp:223-243: u_i = 0 on a contiguous
// range that depends on mpcIter only), u = -(I - W_p E') Hinv f and S u for the three right-hand sides, W_p and S W_p re-strided
constexpr int FMAX_SWEEP = 16;
__global__ __launch_bounds__(256) void sweep_patterns(const double* __restrict__ par, const double* __restrict__ Hinv, size_t smat, const double* __restrict__ hvec,
                                                      const int* __restrict__ e_lo, const int* __restrict__ ne, int npat, int Fmax,
                                                      double* __restrict__ vtab, size_t s_vtab, double* __restrict__ Wt, double* __restrict__ SW, size_t s_W,
                                                      int N, int NG, double dt)
{
    __shared__ double Gm[FMAX_SWEEP][2 * FMAX_SWEEP + 1];
    __shared__ double Wl[256][FMAX_SWEEP + 1];
    __shared__ double hb[3][256], ub[3][256];
    const int p = blockIdx.x, set = blockIdx.y, tid = threadIdx.x;
    const double mass = par[set * PAR + 0];
    const double cs = dt * dt / mass;
    const double* Hs = Hinv + (size_t)set * smat;
    int lo = 0, cnt = 0;
    if (p < npat) { lo = e_lo[p]; cnt = ne[p]; }
    for (int e = tid; e < 3 * N; e += blockDim.x) hb[e / N][e % N] = hvec[((size_t)set * 3 + e / N) * NG + e % N];
    if (cnt > 0) {
        // [G | I] -> [I | G^-1] by Gauss-Jordan without pivoting (G = Hinv[E,E] is symmetric positive definite)
        for (int e = tid; e < cnt * 2 * cnt; e += blockDim.x) {
            const int a = e / (2 * cnt), b = e - a * 2 * cnt;
            Gm[a][b] = b < cnt ? Hs[(size_t)(lo + a) * NG + lo + b] : ((b - cnt == a) ? 1.0 : 0.0);
        }
        __syncthreads();
        for (int pv = 0; pv < cnt; ++pv) {
            const double ip = 1.0 / Gm[pv][pv];
            __syncthreads();
            if (tid < 2 * cnt) Gm[pv][tid] *= ip;
            __syncthreads();
            for (int e = tid; e < cnt * 2 * cnt; e += blockDim.x) {
                const int a = e / (2 * cnt), b = e - a * 2 * cnt;
                if (a != pv && b != pv) Gm[a][b] -= Gm[a][pv] * Gm[pv][b];
            }
            __syncthreads();
            if (tid < cnt && tid != pv) Gm[tid][pv] = 0.0;
            __syncthreads();
        }
        for (int n = tid; n < N; n += blockDim.x)
            for (int e = 0; e < cnt; ++e) {
                double s = 0.0;
                for (int a = 0; a < cnt; ++a) s = fma(Hs[(size_t)n * NG + lo + a], Gm[a][cnt + e], s);
                Wl[n][e] = s;
            }
    }
    __syncthreads();
    for (int e = tid; e < 3 * N; e += blockDim.x) {
        const int b = e / N, n = e - b * N;
        double v = hb[b][n];
        for (int q = 0; q < cnt; ++q) v = fma(-Wl[n][q], hb[b][lo + q], v);
        ub[b][n] = (n >= lo && n < lo + cnt) ? 0.0 : -v;                    // exactly zero on the equality samples
    }
    __syncthreads();
    double* vt = vtab + (size_t)set * s_vtab + (size_t)p * 6 * NT;
    for (int e = tid; e < 3 * N; e += blockDim.x) vt[(size_t)(e / N) * NT + e % N] = ub[e / N][e % N];
    if (tid < 3) {                                                           // S_bar_z u: o[k] = cs sum_{j<k} (k-j) u_j
        DD c1 = {0.0, 0.0}, c2 = {0.0, 0.0};
        for (int k = 0; k < N; ++k) { vt[(size_t)(3 + tid) * NT + k] = cs * (c2.hi + c2.lo); dd_add(c1, ub[tid][k]); dd_add(c2, c1); }
    }
    if (p < npat) {
        double* wt = Wt + (size_t)set * s_W + (size_t)p * Fmax * NT;
        double* sw = SW + (size_t)set * s_W + (size_t)p * Fmax * NT;
        for (int e = tid; e < cnt * N; e += blockDim.x) wt[(size_t)(e / N) * NT + e % N] = Wl[e % N][e / N];
        if (tid < cnt) {
            DD c1 = {0.0, 0.0}, c2 = {0.0, 0.0};
            for (int k = 0; k < N; ++k) { sw[(size_t)tid * NT + k] = cs * (c2.hi + c2.lo); dd_add(c1, Wl[k][tid]); dd_add(c2, c1); }
        }
    }
}

// ---- plans with mid_z != 0 (MPCSolver.cpp:259: f_z carries q_p S' mid_z[idx : idx + N]): per frame idx the offset of the unconstrained
// minimiser dU(idx) = Hinv q_p S' w = q_p sum_k w_k (Hinv S_k'), w = the window -- a combination of the rows of the fallback table HSt,
// which this set already has -- and S dU(idx) (ismpc_tables.cpp "if (!t.flat)").  One workgroup per (frame, set); frames whose window is
// all zero keep the zero the slab was cleared to.
__global__ __launch_bounds__(256) void sweep_du(const double* __restrict__ par, const double* __restrict__ midz, int nmid, const double* __restrict__ HSt, size_t s_HS,
                                                double* __restrict__ dU, double* __restrict__ SdU, size_t s_dU, int N, double dt)
{
    __shared__ double win[256], du[256];
    __shared__ int any;
    const int idx = blockIdx.x, set = blockIdx.y, tid = threadIdx.x;
    if (idx + 2 * N > nmid) return;
    if (tid == 0) any = 0;
    __syncthreads();
    if (tid < N) { win[tid] = midz[idx + tid]; if (win[tid] != 0.0) any = 1; }
    __syncthreads();
    if (!any) return;
    const double mass = par[set * PAR + 0], q_p = par[set * PAR + 1];
    const double cs = dt * dt / mass;
    const double* hs = HSt + (size_t)set * s_HS;
    if (tid < N) {
        double acc = 0.0;
        for (int k = 0; k < N; ++k) acc = fma(win[k], hs[(size_t)k * NT + tid], acc);
        du[tid] = q_p * acc;
        dU[(size_t)set * s_dU + (size_t)idx * NT + tid] = du[tid];
    }
    __syncthreads();
    if (tid == 0) {
        DD c1 = {0.0, 0.0}, c2 = {0.0, 0.0};
        double* o = SdU + (size_t)set * s_dU + (size_t)idx * NT;
        for (int k = 0; k < N; ++k) { o[k] = cs * (c2.hi + c2.lo); dd_add(c1, du[k]); dd_add(c2, c1); }
    }
}

// ---- vtab -> vqT (lane-contiguous copy for the lane-group kernels; ismpc_hip.hip)
__global__ __launch_bounds__(256) void sweep_layout(const double* __restrict__ vtab, size_t s_vtab, double* __restrict__ vqT, size_t s_vqT, int lpi, int R)
{
    const int p = blockIdx.x, set = blockIdx.y;
    const double* vt = vtab + (size_t)set * s_vtab + (size_t)p * 6 * NT;
    double* qt = vqT + (size_t)set * s_vqT + (size_t)p * R * 3 * lpi * 2;
    for (int e = threadIdx.x; e < R * 3 * lpi * 2; e += blockDim.x) {
        const int half = e & 1, li = (e >> 1) % lpi, k = ((e >> 1) / lpi) % 3, r = (e >> 1) / (lpi * 3);
        const int n = li * R + r;                                            // < NT
        qt[e] = vt[(size_t)(2 * k + half) * NT + n];
    }
}

// ---- anticipative tails (MPCSolver.cpp:183-184, 381-383) for this set's eta = sqrt(g / h_des)
__global__ __launch_bounds__(256) void sweep_tail(const double* __restrict__ par, const double* __restrict__ midx, const double* __restrict__ midy, int nmid,
                                                  double* __restrict__ tailx, double* __restrict__ taily, size_t s_tail, int N, double dt)
{
    __shared__ double wgt[256];
    const int set = blockIdx.y;
    const double eta = par[set * PAR + 5];
    if ((int)threadIdx.x < N) wgt[threadIdx.x] = eta * dt * exp(-dt * eta * (double)threadIdx.x);
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nmid) return;
    double sx = 0.0, sy = 0.0;
    if (idx + 2 * N <= nmid)
        for (int i = 0; i < N; ++i) { sx += wgt[i] * midx[idx + N + i]; sy += wgt[i] * midy[idx + N + i]; }
    tailx[(size_t)set * s_tail + idx] = sx; taily[(size_t)set * s_tail + idx] = sy;
}

// ---- did the build work?  Per set: max |I - H X| over the N x N block (a Hessian that is not positive definite, or one so badly
// conditioned that the iteration count ran out, leaves a residual of order one or NaN) and finiteness of the affine tables.
__global__ __launch_bounds__(256) void sweep_check(const double* __restrict__ H, const double* __restrict__ X, size_t smat, const double* __restrict__ vtab, size_t s_vtab,
                                                   int N, int NG, double* __restrict__ resid)
{
    __shared__ double red[256];
    const int set = blockIdx.x, tid = threadIdx.x;
    const double* Hs = H + (size_t)set * smat; const double* Xs = X + (size_t)set * smat;
    double worst = 0.0;
    for (int e = tid; e < N * N; e += blockDim.x) {
        const int i = e / N, j = e - i * N;
        double acc = (i == j) ? -1.0 : 0.0;
        for (int k = 0; k < N; ++k) acc = fma(Hs[(size_t)i * NG + k], Xs[(size_t)k * NG + j], acc);
        const double a = fabs(acc);
        worst = (a > worst || a != a) ? (a != a ? INFINITY : a) : worst;
    }
    const double* vt = vtab + (size_t)set * s_vtab;
    for (size_t e = tid; e < s_vtab; e += blockDim.x) { const double v = vt[e]; if (!(fabs(v) < INFINITY)) worst = INFINITY; }
    red[tid] = worst;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmax(red[tid], red[tid + o]); __syncthreads(); }
    if (tid == 0) resid[set] = red[0];
}

#define SW_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return ISMPC_E_NO_DEVICE; } } while (0)

int dalloc(double** p, size_t n, std::vector<void*>& allocs, std::string& err)
{
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); err = "sweep table allocation failed"; return ISMPC_E_ALLOC; }
    allocs.push_back(q); *p = static_cast<double*>(q);
    return ISMPC_OK;
}

}  // namespace

int sweep_build(const ismpc_params* sets, int K, const Tables& t0, const double* midx_dev, const double* midy_dev, const double* midz_dev, const int* e_lo_dev,
                const int* ne_dev, int lpi, int R, int lpi2, int R2, hipStream_t s, SweepSlabs& o, std::vector<void*>& allocs, std::string& err)
{
    const int N = t0.p.N;
    if (K < 1) { err = "a sweep needs at least one parameter set"; return ISMPC_E_INVALID; }
    if (N > 256) { err = "parameter sweeps cover horizons N <= 256"; return ISMPC_E_UNSUPPORTED; }       // (beyond 128: one instance per wavefront, ismpc_tick_affine<R, true>)
    if (t0.Fmax > FMAX_SWEEP) { err = "parameter sweeps cover F <= 16 double-support samples"; return ISMPC_E_UNSUPPORTED; }
    const int NG = (N + 63) / 64 * 64;
    const double dt = t0.p.mpc_dt;
    o.K = K; o.NG = NG;
    o.s_mat = (size_t)NG * NG; o.s_vtab = (size_t)(t0.npat + 1) * 6 * NT; o.s_vqT = (size_t)(t0.npat + 1) * R * 3 * lpi * 2;
    o.s_vqT2 = lpi2 > 0 ? (size_t)(t0.npat + 1) * R2 * 3 * lpi2 * 2 : 0;
    o.s_W = (size_t)t0.npat * t0.Fmax * NT; o.s_HS = (size_t)N * NT; o.s_tail = (size_t)t0.nmid;
    // per-set scalars; the Newton start 1 / bound(|H|_inf) and the iteration count from cond(H) <= bound / min(q_u, 1)
    std::vector<double> par((size_t)K * PAR);
    int iters = 0;
    for (int k = 0; k < K; ++k) {
        const ismpc_params& p = sets[k];
        const double cs = dt * dt / p.mass, cv = dt / p.mass, n = (double)N;
        const double bound = n * (p.q_p * cs * cs * n * n * n / 3.0 + p.q_v * cv * cv * n) + p.q_u;      // >= max row sum of H
        par[(size_t)k * PAR + 0] = p.mass; par[(size_t)k * PAR + 1] = p.q_p; par[(size_t)k * PAR + 2] = p.q_u; par[(size_t)k * PAR + 3] = p.q_v;
        par[(size_t)k * PAR + 4] = p.h_des; par[(size_t)k * PAR + 5] = std::sqrt(p.g / p.h_des); par[(size_t)k * PAR + 6] = 1.0 / std::max(bound, 1.0);
        par[(size_t)k * PAR + 7] = p.g;
        iters = std::max(iters, (int)std::ceil(std::log2(std::max(bound, 1.0) / std::min(p.q_u, 1.0))) + 8);
    }
    iters = std::min(iters, 96);
    int rc;
    if ((rc = dalloc(&o.par, par.size(), allocs, err))) return rc;
    SW_TRY(hipMemcpyAsync(o.par, par.data(), par.size() * sizeof(double), hipMemcpyHostToDevice, s));
    double** mats[] = { &o.H, &o.X0, &o.X1, &o.T, &o.M1 };
    for (double** m : mats) if ((rc = dalloc(m, (size_t)K * o.s_mat, allocs, err))) return rc;
    if ((rc = dalloc(&o.U, o.s_mat, allocs, err)) || (rc = dalloc(&o.Ut, o.s_mat, allocs, err))) return rc;
    if ((rc = dalloc(&o.hvec, (size_t)K * 3 * NG, allocs, err))) return rc;
    if ((rc = dalloc(&o.vtab, (size_t)K * o.s_vtab, allocs, err)) ||
        (rc = dalloc(&o.vqT, (size_t)K * o.s_vqT, allocs, err)) || (lpi2 > 0 && (rc = dalloc(&o.vqT2, (size_t)K * o.s_vqT2, allocs, err))) ||
        (rc = dalloc(&o.Wt, (size_t)K * o.s_W, allocs, err)) ||
        (rc = dalloc(&o.SW, (size_t)K * o.s_W, allocs, err)) || (rc = dalloc(&o.HSt, (size_t)K * o.s_HS, allocs, err)) ||
        (rc = dalloc(&o.SHSt, (size_t)K * o.s_HS, allocs, err)) || (rc = dalloc(&o.tailx, (size_t)K * o.s_tail, allocs, err)) ||
        (rc = dalloc(&o.taily, (size_t)K * o.s_tail, allocs, err))) return rc;
    SW_TRY(hipMemsetAsync(o.vtab, 0, (size_t)K * o.s_vtab * sizeof(double), s));
    SW_TRY(hipMemsetAsync(o.Wt, 0, (size_t)K * o.s_W * sizeof(double), s));
    SW_TRY(hipMemsetAsync(o.SW, 0, (size_t)K * o.s_W * sizeof(double), s));
    SW_TRY(hipMemsetAsync(o.HSt, 0, (size_t)K * o.s_HS * sizeof(double), s));
    SW_TRY(hipMemsetAsync(o.SHSt, 0, (size_t)K * o.s_HS * sizeof(double), s));
    if (!t0.flat) {
        o.s_dU = (size_t)t0.nmid * NT;
        if ((rc = dalloc(&o.dU, (size_t)K * o.s_dU, allocs, err)) || (rc = dalloc(&o.SdU, (size_t)K * o.s_dU, allocs, err))) return rc;
        SW_TRY(hipMemsetAsync(o.dU, 0, (size_t)K * o.s_dU * sizeof(double), s));
        SW_TRY(hipMemsetAsync(o.SdU, 0, (size_t)K * o.s_dU * sizeof(double), s));
    }
    struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;
    SW_TRY(hipEventCreate(&ev.a)); SW_TRY(hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    SW_TRY(hipEventRecord(e0, s));
    hipLaunchKernelGGL(sweep_init, dim3(K), dim3(256), 0, s, (const double*)o.par, o.H, o.X0, o.s_mat, N, NG, dt);
    hipLaunchKernelGGL(sweep_unit_s, dim3(64), dim3(256), 0, s, o.U, o.Ut, N, NG, dt);
    const dim3 ggrid(NG / 64, NG / 64, K);
    double* X = o.X0; double* Xn = o.X1;
    int launches = 0;
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL((sweep_gemm<0>), ggrid, dim3(256), 0, s, (const double*)o.H, o.s_mat, (const double*)X, o.s_mat, o.T, o.s_mat, (const double*)o.par, 0, NG, N);
        hipLaunchKernelGGL((sweep_gemm<1>), ggrid, dim3(256), 0, s, (const double*)X, o.s_mat, (const double*)o.T, o.s_mat, Xn, o.s_mat, (const double*)o.par, 0, NG, N);
        std::swap(X, Xn); launches += 2;
    }
    if (X != o.X0) SW_TRY(hipMemcpyAsync(o.X0, X, (size_t)K * o.s_mat * sizeof(double), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(sweep_symmetrise, dim3(16, K), dim3(256), 0, s, o.X0, o.s_mat, N, NG);
    // inequality fallback: hs_k = Hinv S_k', shs_k = S hs_k (ismpc_tables.cpp, last block)
    hipLaunchKernelGGL((sweep_gemm<1>), ggrid, dim3(256), 0, s, (const double*)o.X0, o.s_mat, (const double*)o.U, (size_t)0, o.M1, o.s_mat, (const double*)o.par, 1, NG, N);
    // one step of iterative refinement on H M1 = alpha U (as for the right-hand sides in sweep_hvec): two more MFMA products
    hipLaunchKernelGGL((sweep_gemm<1>), ggrid, dim3(256), 0, s, (const double*)o.H, o.s_mat, (const double*)o.M1, o.s_mat, o.T, o.s_mat, (const double*)o.par, 0, NG, N);
    hipLaunchKernelGGL(sweep_residual, dim3(16, K), dim3(256), 0, s, (const double*)o.par, (const double*)o.U, o.T, o.s_mat, NG);
    hipLaunchKernelGGL((sweep_gemm<1>), ggrid, dim3(256), 0, s, (const double*)o.X0, o.s_mat, (const double*)o.T, o.s_mat, o.X1, o.s_mat, (const double*)o.par, 0, NG, N);
    hipLaunchKernelGGL(sweep_accumulate, dim3(16, K), dim3(256), 0, s, o.M1, (const double*)o.X1, o.s_mat, NG);
    hipLaunchKernelGGL((sweep_gemm<2>), ggrid, dim3(256), 0, s, (const double*)o.Ut, (size_t)0, (const double*)o.M1, o.s_mat, o.SHSt, o.s_HS, (const double*)o.par, 1, NG, N);
    launches += 4;
    hipLaunchKernelGGL(sweep_transpose_nt, dim3(16, K), dim3(256), 0, s, (const double*)o.M1, o.s_mat, o.HSt, o.s_HS, N, NG);
    if (!t0.flat)
        hipLaunchKernelGGL(sweep_du, dim3(t0.nmid, K), dim3(256), 0, s, (const double*)o.par, midz_dev, t0.nmid, (const double*)o.HSt, o.s_HS, o.dU, o.SdU, o.s_dU, N, dt);
    hipLaunchKernelGGL(sweep_hvec, dim3(K), dim3(256), 0, s, (const double*)o.par, (const double*)o.X0, (const double*)o.H, o.s_mat, o.hvec, N, NG, dt);
    hipLaunchKernelGGL(sweep_patterns, dim3(t0.npat + 1, K), dim3(256), 0, s, (const double*)o.par, (const double*)o.X0, o.s_mat, (const double*)o.hvec,
                       e_lo_dev, ne_dev, t0.npat, t0.Fmax, o.vtab, o.s_vtab, o.Wt, o.SW, o.s_W, N, NG, dt);
    hipLaunchKernelGGL(sweep_layout, dim3(t0.npat + 1, K), dim3(256), 0, s, (const double*)o.vtab, o.s_vtab, o.vqT, o.s_vqT, lpi, R);
    if (lpi2 > 0) hipLaunchKernelGGL(sweep_layout, dim3(t0.npat + 1, K), dim3(256), 0, s, (const double*)o.vtab, o.s_vtab, o.vqT2, o.s_vqT2, lpi2, R2);
    hipLaunchKernelGGL(sweep_tail, dim3((t0.nmid + 255) / 256, K), dim3(256), 0, s, (const double*)o.par, midx_dev, midy_dev, t0.nmid, o.tailx, o.taily, o.s_tail, N, dt);
    SW_TRY(hipGetLastError());
    SW_TRY(hipEventRecord(e1, s));
    // the build is checked, not trusted: |I - H X| of every set and finite tables
    double* resid_dev = nullptr;
    if ((rc = dalloc(&resid_dev, (size_t)K, allocs, err))) return rc;
    hipLaunchKernelGGL(sweep_check, dim3(K), dim3(256), 0, s, (const double*)o.H, (const double*)o.X0, o.s_mat, (const double*)o.vtab, o.s_vtab, N, NG, resid_dev);
    std::vector<double> resid((size_t)K);
    SW_TRY(hipMemcpyAsync(resid.data(), resid_dev, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, s));
    SW_TRY(hipStreamSynchronize(s));
    SW_TRY(hipEventElapsedTime(&o.build_ms, e0, e1));
    o.max_residual = 0.0;
    for (int k = 0; k < K; ++k) {
        // a converged iterate leaves cond(H) x (relative error of X ~ cond eps): far below 1e-3 for any set the affine tables make sense for; an
        // iteration that ran out of budget, or a Hessian that is not positive definite, leaves O(1) or NaN.  (Accuracy is cond(H) eps, as for
        // any inverse in fp64: ismpc_sweep_verify_tables measures it against the long-double host build.)
        if (!(resid[k] <= 1e-3)) {
            err = "parameter set " + std::to_string(k) + ": the vertical Hessian could not be inverted (not positive definite, or conditioned beyond the "
                  "Newton-Schulz iteration budget): residual |I - H X| = " + std::to_string(resid[k]);
            return ISMPC_E_NUMERIC;
        }
        o.max_residual = std::max(o.max_residual, resid[k]);
    }
    o.newton_iters = iters; o.gemm_launches = launches;
    return ISMPC_OK;
}

}  // namespace ismpc
