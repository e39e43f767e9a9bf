// One-off host precompute for the batched ISMPC tick (see ismpc_tables.hpp).
// Follows the reference's MPCSolver constructor (AMR_code_DART/MPCSolver.cpp:124-198)
// for WHAT is computed; HOW is different: closed forms of the Toeplitz blocks,
// long-double Cholesky inverse of the constant vertical Hessian, and one
// Schur-complement correction per mpcIter pattern of the u_i = 0 equalities.
#include "ismpc_tables.hpp"
#include <cmath>
#include <algorithm>

namespace ismpc {

typedef long double ld;

// In-place lower Cholesky of an n x n row-major SPD matrix (long double).
static bool cholesky(std::vector<ld>& a, int n)
{
    for (int j = 0; j < n; ++j) {
        ld s = a[(size_t)j*n+j];
        for (int k = 0; k < j; ++k) s -= a[(size_t)j*n+k]*a[(size_t)j*n+k];
        if (!(s > 0)) return false;
        ld ljj = sqrtl(s);
        a[(size_t)j*n+j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            ld t = a[(size_t)i*n+j];
            for (int k = 0; k < j; ++k) t -= a[(size_t)i*n+k]*a[(size_t)j*n+k];
            a[(size_t)i*n+j] = t / ljj;
        }
    }
    return true;
}

// inv = (L L')^-1 given lower L.
static void chol_inverse(const std::vector<ld>& L, int n, std::vector<ld>& inv)
{
    std::vector<ld> Li((size_t)n*n, 0);          // L^-1, lower
    for (int j = 0; j < n; ++j) {
        Li[(size_t)j*n+j] = 1 / L[(size_t)j*n+j];
        for (int i = j + 1; i < n; ++i) {
            ld s = 0;
            for (int k = j; k < i; ++k) s -= L[(size_t)i*n+k]*Li[(size_t)k*n+j];
            Li[(size_t)i*n+j] = s / L[(size_t)i*n+i];
        }
    }
    inv.assign((size_t)n*n, 0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            ld s = 0;
            for (int k = i; k < n; ++k) s += Li[(size_t)k*n+i]*Li[(size_t)k*n+j];
            inv[(size_t)i*n+j] = s; inv[(size_t)j*n+i] = s;
        }
}

int build_tables(const ismpc_params& p, const double* ftsp, int rows, Tables& t, std::string& err)
{
    if (!ftsp || rows < 2) { err = "footstep plan needs at least 2 rows"; return ISMPC_E_INVALID; }
    if (p.N < 1 || p.N > 256) { err = "horizon N must be in [1, 256]"; return p.N > 256 ? ISMPC_E_UNSUPPORTED : ISMPC_E_INVALID; }
    if (p.S < 0 || p.F < 1 || p.S + p.F > 4096) { err = "S must be >= 0 and F >= 1"; return ISMPC_E_INVALID; }
    if (!(p.mpc_dt > 0) || !(p.control_dt > 0) || !(p.mass > 0) || !(p.g > 0) || !(p.h_des > 0)) {
        err = "mpc_dt, control_dt, mass, g, h_des must be positive"; return ISMPC_E_INVALID;
    }
    if (!(p.q_u > 0) || p.q_p < 0 || p.q_v < 0) { err = "weights: q_u > 0, q_p >= 0, q_v >= 0"; return ISMPC_E_INVALID; }
    if ((int)(100 * p.mpc_dt) <= 0) { err = "(int)(100*mpc_dt) is 0: the reference's tick gate (MPCSolver.cpp:214) divides by it"; return ISMPC_E_INVALID; }

    t.p = p;
    t.eta = std::sqrt(p.g / p.h_des);
    t.rows = rows;
    const int N = p.N, S = p.S, F = p.F;
    t.nmid = rows * (S + F);
    t.NP = (N + 15) / 16 * 16;
    t.npat = S + F;
    t.Fmax = F;
    t.tick_divisor = (int)(100 * p.mpc_dt);
    const ld dt = p.mpc_dt, m = p.mass;

    // ---- vertical Hessian, MPCSolver.cpp:258, from the closed forms of :144-154:
    // S_bar_z(k,j) = (k-j) dt^2/m, S_bar_z_v(k,j) = dt/m for j < k, else 0.
    std::vector<ld> H((size_t)N*N);
    const ld cs = dt*dt/m, cv = dt/m;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j) {
            ld a = 0;
            for (int k = i + 1; k < N; ++k) a += (ld)(k - i) * (ld)(k - j);
            ld b = (ld)(N - 1 - i);
            ld v = (ld)p.q_p * cs*cs * a + (ld)p.q_v * cv*cv * b + (i == j ? (ld)p.q_u : 0);
            H[(size_t)i*N+j] = v; H[(size_t)j*N+i] = v;
        }
    std::vector<ld> L = H;
    if (!cholesky(L, N)) { err = "vertical Hessian is not positive definite"; return ISMPC_E_NUMERIC; }
    std::vector<ld> Hinv;
    chol_inverse(L, N, Hinv);
    t.Hinv.assign((size_t)t.NP*t.NP, 0.0);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) t.Hinv[(size_t)i*t.NP+j] = (double)Hinv[(size_t)i*N+j];

    // ---- equality patterns, MPCSolver.cpp:223-243: u_i = 0 on a contiguous range that
    // depends only on mpcIter.  Correction  u = u_unc - W (u_unc)_E,  W = Hinv[:,E] (Hinv[E,E])^-1.
    t.e_lo.assign(t.npat, 0); t.ne.assign(t.npat, 0);
    t.W.assign((size_t)t.npat * t.Fmax * t.NP, 0.0);
    std::vector<ld> Wld((size_t)t.npat * t.Fmax * N, 0);     // long-double copy for the affine tables below
    for (int it = 0; it < t.npat; ++it) {
        // the reference fills Aeq_z inside `for (i = 0; i < N; i++)` (:231): row i-S, column i-mpcIter for
        // S <= i < min(S+F, N) while mpcIter < S, else column i for i < min(S+F-mpcIter, N)
        int lo, hi;
        if (it < S) { lo = S - it; hi = std::min(S + F, N) - it; } else { lo = 0; hi = std::min(S + F - it, N); }
        if (hi <= lo) { lo = 0; hi = 0; }
        int cnt = hi - lo;
        t.e_lo[it] = lo; t.ne[it] = cnt;
        if (cnt == 0) continue;
        std::vector<ld> G((size_t)cnt*cnt), Gi;
        for (int a = 0; a < cnt; ++a)
            for (int b = 0; b < cnt; ++b) G[(size_t)a*cnt+b] = Hinv[(size_t)(lo+a)*N + (lo+b)];
        if (!cholesky(G, cnt)) { err = "equality Schur complement is not positive definite"; return ISMPC_E_NUMERIC; }
        chol_inverse(G, cnt, Gi);
        for (int e = 0; e < cnt; ++e)
            for (int n = 0; n < N; ++n) {
                ld s = 0;
                for (int a = 0; a < cnt; ++a) s += Hinv[(size_t)n*N + (lo+a)] * Gi[(size_t)a*cnt+e];
                t.W[((size_t)it*t.Fmax + e)*t.NP + n] = (double)s;
                Wld[((size_t)it*t.Fmax + e)*N + n] = s;
            }
    }

    // ---- ftsp_midpoint, MPCSolver.cpp:167-180 (same operation order as the reference)
    t.midx.assign(t.nmid, 0.0); t.midy.assign(t.nmid, 0.0); t.midz.assign(t.nmid, 0.0);
    std::vector<double>* col[3] = { &t.midx, &t.midy, &t.midz };
    for (int i = 0; i < rows - 1; ++i)
        for (int c = 0; c < 3; ++c) {
            const double a = ftsp[i*4+c], b = ftsp[(i+1)*4+c];
            for (int r = 0; r < S; ++r) (*col[c])[i*(S+F)+r] = a * 1.0;
            for (int r = 0; r < F; ++r) (*col[c])[i*(S+F)+S+r] = a * 1.0 + (b - a) * ((double)r / (double)F);
        }
    t.ftsp_t.resize(rows);
    for (int i = 0; i < rows; ++i) t.ftsp_t[i] = ftsp[i*4+3];

    // ---- anticipative tail, MPCSolver.cpp:183-184 and :381-383: depends on idx only
    std::vector<double> deltas(N);
    for (int i = 0; i < N; ++i) deltas[i] = std::exp(-p.mpc_dt * t.eta * i);
    t.tailx.assign(t.nmid, 0.0); t.taily.assign(t.nmid, 0.0);
    for (int idx = 0; idx + 2*N <= t.nmid; ++idx) {
        double sx = 0, sy = 0;
        for (int i = 0; i < N; ++i) {
            const double wgt = t.eta * p.mpc_dt * deltas[i];
            sx += wgt * t.midx[idx+N+i]; sy += wgt * t.midy[idx+N+i];
        }
        t.tailx[idx] = sx; t.taily[idx] = sy;
    }

    // ---- affine tables of the vertical stage (see ismpc_tables.hpp) ----
    const int NT = Tables::NT;
    auto apply_St = [&](const std::vector<ld>& r, std::vector<ld>& o, bool vel) {   // S_bar_z' r  or  S_bar_z_v' r
        o.assign(N, 0);
        if (vel) { ld acc = 0; for (int i = N - 1; i >= 0; --i) { o[i] = cv * acc; acc += r[i]; } }
        else { ld t1 = 0, t2 = 0; for (int i = N - 1; i >= 0; --i) { o[i] = cs * t2; t1 += r[i]; t2 += t1; } }
    };
    auto apply_S = [&](const std::vector<ld>& u, std::vector<ld>& o) {              // S_bar_z u
        o.assign(N, 0); ld c1 = 0, c2 = 0;
        for (int k = 0; k < N; ++k) { o[k] = cs * c2; c1 += u[k]; c2 += c1; }
    };
    auto apply_Hinv = [&](const std::vector<ld>& f, std::vector<ld>& o) {
        o.assign(N, 0);
        for (int i = 0; i < N; ++i) { ld s = 0; for (int j = 0; j < N; ++j) s += Hinv[(size_t)i*N+j] * f[j]; o[i] = s; }
    };
    // f = f0 + z fa + zdot fb - q_p S' mid_z(window)      (MPCSolver.cpp:259)
    std::vector<ld> one(N, 1), tvec(N), tgv(N), tgz(N), tmp, tmp2, f0(N), fa(N), fb(N);
    t.tz.assign(NT, 0.0); t.tg.assign(NT, 0.0);
    for (int k = 0; k < N; ++k) {
        tvec[k] = (ld)(k + 1) * dt;                               // T_bar_z(k,1)
        tgz[k] = -(ld)p.g * dt * dt * ((ld)k * (ld)(k + 1) / 2);  // T_bar_g_z(k)
        tgv[k] = -(ld)p.g * dt * (ld)k;                           // T_bar_g_z_v(k)
        t.tz[k] = (double)tvec[k]; t.tg[k] = (double)tgz[k];
    }
    {
        std::vector<ld> r0(N);
        for (int k = 0; k < N; ++k) r0[k] = tgz[k] - (ld)p.h_des;
        apply_St(r0, tmp, false); apply_St(tgv, tmp2, true);
        for (int i = 0; i < N; ++i) f0[i] = (ld)p.q_p * tmp[i] + (ld)p.q_v * tmp2[i] - (ld)p.q_u * m * (ld)p.g;
        apply_St(one, tmp, false);
        for (int i = 0; i < N; ++i) fa[i] = (ld)p.q_p * tmp[i];
        apply_St(tvec, tmp, false); apply_St(one, tmp2, true);
        for (int i = 0; i < N; ++i) fb[i] = (ld)p.q_p * tmp[i] + (ld)p.q_v * tmp2[i];
    }
    std::vector<ld> h0, ha, hb;
    apply_Hinv(f0, h0); apply_Hinv(fa, ha); apply_Hinv(fb, hb);
    t.vtab.assign((size_t)(t.npat + 1) * 6 * NT, 0.0);
    for (int pat = 0; pat <= t.npat; ++pat) {
        const int lo = pat < t.npat ? t.e_lo[pat] : 0, cnt = pat < t.npat ? t.ne[pat] : 0;
        const std::vector<ld>* src[3] = { &h0, &ha, &hb };
        for (int b = 0; b < 3; ++b) {
            std::vector<ld> u(N), su;
            for (int n = 0; n < N; ++n) {
                ld v = (*src[b])[n];
                for (int e = 0; e < cnt; ++e) v -= Wld[((size_t)pat*t.Fmax + e)*N + n] * (*src[b])[lo + e];
                u[n] = -v;                                          // u = -P_p f
            }
            for (int e = 0; e < cnt; ++e) u[lo + e] = 0;            // exactly zero on the equality samples
            apply_S(u, su);
            for (int n = 0; n < N; ++n) {
                t.vtab[((size_t)pat*6 + b)*NT + n] = (double)u[n];
                t.vtab[((size_t)pat*6 + 3 + b)*NT + n] = (double)su[n];
            }
        }
    }
    t.flat = true;
    for (int i = 0; i < t.nmid; ++i) if (t.midz[i] != 0.0) { t.flat = false; break; }
    if (!t.flat) {
        t.dU.assign((size_t)t.nmid * NT, 0.0); t.SdU.assign((size_t)t.nmid * NT, 0.0);
        std::vector<ld> win(N), g, du, sdu;
        for (int idx = 0; idx + 2*N <= t.nmid; ++idx) {
            bool any = false;
            for (int k = 0; k < N; ++k) { win[k] = t.midz[idx + k]; any = any || win[k] != 0; }
            if (!any) continue;
            apply_St(win, g, false);
            for (int i = 0; i < N; ++i) g[i] *= (ld)p.q_p;          // f -= g  ->  u_unc += Hinv g
            apply_Hinv(g, du); apply_S(du, sdu);
            for (int n = 0; n < N; ++n) { t.dU[(size_t)idx*NT + n] = (double)du[n]; t.SdU[(size_t)idx*NT + n] = (double)sdu[n]; }
        }
    }
    // equality corrections re-strided to NT (non-flat plans and the inequality fallback both use them)
    {
        t.Wt.assign((size_t)t.npat * t.Fmax * NT, 0.0); t.SW.assign((size_t)t.npat * t.Fmax * NT, 0.0);
        std::vector<ld> col(N), scol;
        for (int it = 0; it < t.npat; ++it)
            for (int e = 0; e < t.ne[it]; ++e) {
                for (int n = 0; n < N; ++n) col[n] = Wld[((size_t)it*t.Fmax + e)*N + n];
                apply_S(col, scol);
                for (int n = 0; n < N; ++n) { t.Wt[((size_t)it*t.Fmax + e)*NT + n] = (double)col[n]; t.SW[((size_t)it*t.Fmax + e)*NT + n] = (double)scol[n]; }
            }
    }
    // ---- inequality fallback (0 <= S u <= 1e4 active, MPCSolver.cpp:158-160): for row k of S_bar_z,
    // hs_k = Hinv S_k' and shs_k = S hs_k; the equality pattern is folded in at run time with W / SW.
    {
        t.HSt.assign((size_t)N * NT, 0.0); t.SHSt.assign((size_t)N * NT, 0.0);
        std::vector<ld> srow(N), hs, shs;
        for (int k = 0; k < N; ++k) {
            for (int j = 0; j < N; ++j) srow[j] = j < k ? cs * (ld)(k - j) : 0;      // S_bar_z(k, j)
            apply_Hinv(srow, hs); apply_S(hs, shs);
            for (int n = 0; n < N; ++n) { t.HSt[(size_t)k*NT + n] = (double)hs[n]; t.SHSt[(size_t)k*NT + n] = (double)shs[n]; }
        }
    }
    return ISMPC_OK;
}

}  // namespace ismpc
