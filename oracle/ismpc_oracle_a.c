/*
 * ismpc_oracle_a.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C restatement of the reference's "Formulation A" (classic ISMPC with footstep
 * adaptation), i.e. the MATLAB generators that produced every checked-in trajectory fixture
 * under AMR_code_DART/MATLAB_trajectories/.  Followed line by line (paths relative to the
 * reference root):
 *   orc_a_plan        trotting/init_quadruped.m:5-184      (gait 0)
 *                     walking/init_quadruped2.m:5-284      (gait 1)
 *   orc_a_create      walking/quad_walk_no_plots.m:6-110   /  trotting/quad_as_bip_no_plots.m:6-103
 *   orc_a_tick        walking/quad_walk_no_plots.m:127-331,509-559  /  trotting/quad_as_bip_no_plots.m:116-316,436-479
 * 1-based indexing is kept (arrays are allocated one longer) so that every index reads like the
 * .m file.  The swing-foot re-placement QPs (quad_walk_no_plots.m:336-504) only edit foot_plan,
 * never fs_plan / the CoM, and are not restated.
 *
 * The MATLAB code solves ONE quadprog in 2(C+F) variables whose Hessian is block diagonal and
 * whose constraint rows never couple x and y, plus 4C all-zero inequality rows and two all-zero
 * equality rows; here the two axes are solved as two (C+F)-variable QPs in the reference's own
 * qpOASES convention (utils.cpp:89-139: stacked lbA <= A x <= ubA) with the zero rows dropped --
 * same minimiser.  QP backend pluggable exactly as in ismpc_oracle.c.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef int (*orc_qp_fn)(int nV, int nC, const double* H, const double* g,
                         const double* A, const double* lbA, const double* ubA,
                         double* x, int* nWSR);
int orc_qp_gi(int n, int nC, const double* H, const double* g0, const double* A,
              const double* lbA, const double* ubA, double* x, int* nWSR);

typedef struct orc_a_gait {
    int gait;              /* 0 = trot (init_quadruped.m), 1 = walk (init_quadruped2.m) */
    int n_gait;            /* N_gait = 100                                   :5 */
    double disp_A, phi;    /* step length, heading                            :7,9 */
    double disp_B, disp_C; /* 0.259394, 0.88                                  :16-17 */
    double disp_i, disp_o, disp_forw;   /* 0.4, 0.4, 0.5                      :30-34 */
} orc_a_gait;

typedef struct orc_a_params {
    int C, P, F;           /* 100/200/3 (walk), 160/320/3 (trot)   quad_walk_no_plots.m:30-32 */
    int step, ds;          /* step_duration, dsSamples: 50/30 (walk), 80/50 (trot) */
    int n_gait;            /* NF */
    double dt;             /* mpcTimeStep 0.01 */
    double height;         /* 0.56 */
    double grav;           /* 9.8 (NOT 9.81), quad_walk_no_plots.m:35 */
    double w;              /* centroid_size = foot_size = 0.02 */
    double Qf;             /* Qfootsteps: 1e9 (walk), 1e7 (trot) */
    double disp_forw, disp_forw_dummy, disp_L;   /* 0.5, 0.25, (disp_o+disp_i)/2 = 0.4 */
} orc_a_params;

/* instance state: everything the loop carries from one tick to the next */
typedef struct orc_a_state {
    double x, xd, xz, y, yd, yz;
    double cur_x, cur_y;   /* current_xfs, current_yfs */
    double pred_x, pred_y; /* predicted_xfs(1), predicted_yfs(1) of the last tick */
    int fc;                /* fsCounter (1-based) */
    int j;                 /* next tick index (1-based) */
} orc_a_state;

typedef struct orc_a_tick_out {
    double com_before[2];  /* x_store(j), y_store(j): what row j of ComTrajectory holds */
    double vel_after[2];   /* xd_store(j), yd_store(j): row j of ComVelocity */
    double u0[2];          /* predicted_xzd(1), predicted_yzd(1) */
    double f0[2];          /* predicted_xfs(1), predicted_yfs(1) */
    int rv[2], nwsr[2];
    int fc, stepped;       /* fsCounter used by this tick; 1 if the footstep counter advanced after it */
} orc_a_tick_out;

typedef struct orc_a_sim {
    orc_a_params p;
    double eta;
    int ncl;                       /* length of cl_x / cl_y */
    int nplan;                     /* rows of fs_plan */
    double *fsx, *fsy;             /* fs_plan(:,1), fs_plan(:,2), 1-based */
    double *clx, *cly;             /* centreline, 1-based */
    double A_upd[9], B_upd[3];
    orc_a_state st;
    orc_qp_fn qp;
    /* swing-foot re-placement (second quadprog of the scripts): optional, enabled by orc_a_enable_feet */
    double* fp;                    /* foot_plan, (fprows+2) x 8, 1-based rows, columns BL(1,2) BR(3,4) FR(5,6) FL(7,8) */
    int fprows, gait, counter;     /* `counter` of quad_walk_no_plots.m:114 (incremented with fsCounter, never wrapped) */
    double phi, disp_i, disp_o, disp_forw_feet;
} orc_a_sim;

/* ---- helpers ---------------------------------------------------------- */
/* MATLAB linspace(d1, d2, n) */
static void linspace_m(double d1, double d2, int n, double* y)
{
    int n1 = n - 1;
    for (int k = 0; k <= n1; ++k) y[k] = d1 + (k * (d2 - d1)) / n1;
    if (n > 0) { y[0] = d1; y[n1] = d2; }
}

/* polyfit([x1 x2],[y1 y2],1) then the intersection of the two lines (init_quadruped.m:172-183) */
static void diag_intersection(const double* fp /* 8 values, 1-based cols 1..8 at fp[0..7] */, double* cx, double* cy)
{
    double m1 = (fp[5] - fp[1]) / (fp[4] - fp[0]);       /* through (1,2) and (5,6) */
    double b1 = fp[1] - m1 * fp[0];
    double m2 = (fp[7] - fp[3]) / (fp[6] - fp[2]);       /* through (3,4) and (7,8) */
    double b2 = fp[3] - m2 * fp[2];
    double x = (b2 - b1) / (m1 - m2);
    *cx = x; *cy = m1 * x + b1;
}

/* trotting/init_quadruped.m:5-184 and walking/init_quadruped2.m:5-284.
 * foot_plan: (n_gait+2) x 8 row-major, rows 1..n_gait(+1) used, columns BL(1,2) BR(3,4) FR(5,6) FL(7,8).
 * center   : (n_gait+1) x 2, rows 1..n_gait used.  Returns number of foot_plan rows written. */
int orc_a_plan(const orc_a_gait* gp, double* foot_plan, double* center)
{
    const int NG = gp->n_gait;
    const double disp_A = gp->disp_A, phi = gp->phi, disp_B = gp->disp_B, disp_C = gp->disp_C;
    const double disp_i = gp->disp_i, disp_o = gp->disp_o, disp_forw = gp->disp_forw;
    const double disp_forw_dummy = disp_forw / 2;
    const double disp_vertical = disp_i < disp_o ? disp_i : disp_o;
    const double disp_vertical_dummy = disp_vertical / 2;
    double x_passo = disp_A * cos(phi), y_passo = disp_A * sin(phi);
    double x_passo_dummy = disp_A * cos(phi) / 2, y_passo_dummy = disp_A * sin(phi) / 2;
    /* :62-81 */
    if (y_passo_dummy > disp_vertical_dummy || x_passo_dummy > disp_forw_dummy) {
        if (phi > atan(disp_vertical_dummy / disp_forw_dummy)) { y_passo_dummy = disp_vertical_dummy; x_passo_dummy = disp_vertical_dummy * cos(phi) / sin(phi); }
        else { x_passo_dummy = disp_forw_dummy; y_passo_dummy = disp_forw_dummy * sin(phi) / cos(phi); }
    }
    /* :84-102 */
    if (y_passo > disp_vertical || x_passo > disp_forw) {
        if (phi > atan(disp_vertical / disp_forw)) { y_passo = disp_vertical; x_passo = disp_vertical * cos(phi) / sin(phi); }
        else { x_passo = disp_forw; y_passo = disp_forw * sin(phi) / cos(phi); }
    }
    const int rows = NG + 2;
#define FP(r, c) foot_plan[(size_t)(r) * 8 + ((c) - 1)]
    /* columns: BL = 1,2 ; BR = 3,4 ; FR = 5,6 ; FL = 7,8 */
    for (int r = 0; r < rows; ++r) {
        FP(r,1) = 0.0;    FP(r,2) = disp_B;
        FP(r,3) = 0.0;    FP(r,4) = -disp_B;
        FP(r,5) = disp_C; FP(r,6) = -disp_B;
        FP(r,7) = disp_C; FP(r,8) = disp_B;
    }
    int used = NG;
    if (gp->gait == 0) {
        /* trot, init_quadruped.m:113-149 */
        FP(2,1) = x_passo_dummy;          FP(2,5) = disp_C + x_passo_dummy;
        FP(2,2) = disp_B + y_passo_dummy; FP(2,6) = -disp_B + y_passo_dummy;
        for (int j = 3; j <= NG; ++j) {
            if (j % 2 == 0) {
                FP(j,1) = FP(j-1,1) + x_passo; FP(j,5) = FP(j-1,5) + x_passo;
                FP(j,3) = FP(j-1,3);           FP(j,7) = FP(j-1,7);
                FP(j,2) = FP(j-1,2) + y_passo; FP(j,6) = FP(j-1,6) + y_passo;
                FP(j,4) = FP(j-1,4);           FP(j,8) = FP(j-1,8);
            } else {
                FP(j,3) = FP(j-1,3) + x_passo; FP(j,7) = FP(j-1,7) + x_passo;
                FP(j,1) = FP(j-1,1);           FP(j,5) = FP(j-1,5);
                FP(j,4) = FP(j-1,4) + y_passo; FP(j,8) = FP(j-1,8) + y_passo;
                FP(j,2) = FP(j-1,2);           FP(j,6) = FP(j-1,6);
            }
        }
        /* centre, :165-184 */
        for (int r = 0; r <= NG; ++r) { center[r*2] = 0; center[r*2+1] = 0; }
        center[1*2] = disp_C / 2;
        for (int k = 2; k <= NG; ++k) diag_intersection(&FP(k,1), &center[k*2], &center[k*2+1]);
    } else {
        /* walk, init_quadruped2.m:113-217 */
        FP(3,7) = disp_C + x_passo_dummy; FP(4,7) = FP(3,7); FP(5,7) = FP(3,7);
        FP(2,3) = FP(1,3); FP(3,3) = FP(1,3); FP(4,3) = FP(3,3); FP(5,3) = FP(4,3) + x_passo_dummy;
        FP(3,8) = disp_B + y_passo_dummy; FP(4,8) = FP(3,8); FP(5,8) = FP(3,8);
        FP(2,4) = FP(1,4); FP(3,4) = FP(1,4); FP(4,4) = FP(3,4); FP(5,4) = FP(4,4) + y_passo_dummy;
        for (int j = 6; j <= NG; j += 8) {
            for (int c = 0; c < 2; ++c) {                 /* c = 0: x columns, c = 1: y columns */
                const double passo = c == 0 ? x_passo : y_passo;
                const int BL = 1 + c, BR = 3 + c, FR = 5 + c, FL = 7 + c;
                FP(j,FR) = FP(j-1,FR); FP(j+1,FR) = FP(j,FR) + passo;
                for (int k = 2; k <= 7; ++k) FP(j+k,FR) = FP(j+1,FR);
                FP(j,BL) = FP(j-1,BL); FP(j+1,BL) = FP(j,BL); FP(j+2,BL) = FP(j,BL);
                FP(j+3,BL) = FP(j+2,BL) + passo;
                for (int k = 4; k <= 7; ++k) FP(j+k,BL) = FP(j+3,BL);
                FP(j,FL) = FP(j-1,FL);
                for (int k = 1; k <= 4; ++k) FP(j+k,FL) = FP(j,FL);
                FP(j+5,FL) = FP(j+4,FL) + passo; FP(j+6,FL) = FP(j+5,FL); FP(j+7,FL) = FP(j+5,FL);
                FP(j,BR) = FP(j-1,BR);
                for (int k = 1; k <= 6; ++k) FP(j+k,BR) = FP(j,BR);
                FP(j+7,BR) = FP(j+6,BR) + passo;
            }
            if (j + 7 > used) used = j + 7;               /* the arrays auto-grow to row 101 */
        }
        /* centre, :236-284: quadruple-support rows from the diagonals, triple-support rows hold */
        for (int r = 0; r <= NG; ++r) { center[r*2] = 0; center[r*2+1] = 0; }
        center[1*2] = disp_C / 2;
        for (int j = 1; j <= NG - 4; j += 8) {
            for (int k = 0; k <= 6; k += 2) diag_intersection(&FP(j+k,1), &center[(j+k)*2], &center[(j+k)*2+1]);
            for (int k = 1; k <= 7; k += 2) { center[(j+k)*2] = center[(j+k-1)*2]; center[(j+k)*2+1] = center[(j+k-1)*2+1]; }
        }
    }
#undef FP
    return used;
}

/* ---- simulation ------------------------------------------------------- */
static int fs_timing(const orc_a_sim* s, int k) { return s->p.step * (k - 1); }   /* fs_timing(k), 1-based */

/* quad_walk_no_plots.m:86-99 (initial = 1) and :540-549 (initial = 0) */
static void build_centerline(orc_a_sim* s, int initial)
{
    const int step = s->p.step, ds = s->p.ds, NF = s->p.n_gait;
    double* lin = (double*)malloc(sizeof(double) * (ds > 0 ? ds : 1));
    for (int axis = 0; axis < 2; ++axis) {
        const double* fs = axis == 0 ? s->fsx : s->fsy;
        double* cl = axis == 0 ? s->clx : s->cly;
        int n = 0;
        if (initial) {
            for (int k = 0; k < step - ds; ++k) cl[++n] = fs[1] * 1.0;
            linspace_m(fs[1], fs[2], ds, lin);
            for (int k = 0; k < ds; ++k) cl[++n] = lin[k];
        } else {
            for (int k = 0; k < step; ++k) cl[++n] = fs[1] * 1.0;
        }
        for (int i = 2; i <= NF - 1; ++i) {
            for (int k = 0; k < step - ds; ++k) cl[++n] = fs[i] * 1.0;
            linspace_m(fs[i], fs[i+1], ds, lin);
            for (int k = 0; k < ds; ++k) cl[++n] = lin[k];
        }
        s->ncl = n;
    }
    free(lin);
}

void orc_a_destroy(orc_a_sim* s)
{
    if (!s) return;
    free(s->fsx); free(s->fsy); free(s->clx); free(s->cly); free(s->fp); free(s);
}

/* center: (n_gait+1) x 2, rows 1..n_gait (as orc_a_plan writes it) */
orc_a_sim* orc_a_create(const orc_a_params* p, const double* center, double disp_C)
{
    orc_a_sim* s = (orc_a_sim*)calloc(1, sizeof(*s));
    s->p = *p;
    s->eta = sqrt(p->grav / p->height);
    s->nplan = p->n_gait;
    s->fsx = (double*)calloc(p->n_gait + 2, 8); s->fsy = (double*)calloc(p->n_gait + 2, 8);
    for (int i = 1; i <= p->n_gait; ++i) { s->fsx[i] = center[i*2]; s->fsy[i] = center[i*2+1]; }   /* fs_plan = center */
    const int maxcl = (p->n_gait + 1) * p->step + 8;
    s->clx = (double*)calloc(maxcl, 8); s->cly = (double*)calloc(maxcl, 8);
    build_centerline(s, 1);
    const double dt = p->dt, eta = s->eta;
    const double ch = cosh(eta * dt), sh = sinh(eta * dt);
    const double A[9] = { ch, sh/eta, 1-ch, eta*sh, ch, -eta*sh, 0, 0, 1 };
    const double B[3] = { dt - sh/eta, 1-ch, dt };
    memcpy(s->A_upd, A, sizeof(A)); memcpy(s->B_upd, B, sizeof(B));
    s->st.x = disp_C / 2; s->st.xd = 0; s->st.xz = disp_C / 2;
    s->st.y = 0; s->st.yd = 0; s->st.yz = 0;
    s->st.cur_x = s->fsx[1]; s->st.cur_y = s->fsy[1];
    s->st.pred_x = 0; s->st.pred_y = 0;
    s->st.fc = 1; s->st.j = 1;
    s->qp = orc_qp_gi;
    return s;
}

void orc_a_set_qp_backend(orc_a_sim* s, orc_qp_fn fn) { s->qp = fn ? fn : orc_qp_gi; }
void orc_a_get_state(const orc_a_sim* s, orc_a_state* st) { *st = s->st; }
void orc_a_set_state(orc_a_sim* s, const orc_a_state* st) { s->st = *st; }
int  orc_a_plan_rows(const orc_a_sim* s) { return s->nplan; }
int  orc_a_cl_len(const orc_a_sim* s) { return s->ncl; }
/* copies fs_plan (rows 1..n) and the centreline (1..ncl) out, 0-based in the destination */
void orc_a_get_plan(const orc_a_sim* s, double* fsx, double* fsy, double* clx, double* cly)
{
    if (fsx) memcpy(fsx, s->fsx + 1, 8 * s->nplan);
    if (fsy) memcpy(fsy, s->fsy + 1, 8 * s->nplan);
    if (clx) memcpy(clx, s->clx + 1, 8 * s->ncl);
    if (cly) memcpy(cly, s->cly + 1, 8 * s->ncl);
}
void orc_a_set_plan(orc_a_sim* s, const double* fsx, const double* fsy, const double* clx, const double* cly, int ncl)
{
    memcpy(s->fsx + 1, fsx, 8 * s->nplan); memcpy(s->fsy + 1, fsy, 8 * s->nplan);
    memcpy(s->clx + 1, clx, 8 * ncl); memcpy(s->cly + 1, cly, 8 * ncl); s->ncl = ncl;
}

/* Puts the simulation into the situation a mid-run tick sees, from the compact per-instance record the batched
 * generators carry (base plan + one plan shift per axis, quad_walk_no_plots.m:535-536; which centreline structure is
 * live, :86-99 or :540-549): fs_plan = center + off, centreline rebuilt from it.  center as orc_a_create takes it. */
void orc_a_load_shifted(orc_a_sim* s, const orc_a_state* st, const double* center, double off_x, double off_y, int rebuilt)
{
    s->st = *st;
    for (int i = 1; i <= s->p.n_gait; ++i) { s->fsx[i] = center[i*2] + off_x; s->fsy[i] = center[i*2+1] + off_y; }
    build_centerline(s, rebuilt ? 0 : 1);
}

/* One axis of the tick's QP: quad_walk_no_plots.m:153-293 restricted to one coordinate.
 * sol: C+F values (zmp velocities then footsteps). */
static int solve_axis(orc_a_sim* s, int axis, const double* mapping /* C x (F+1) */, double* sol, int* nwsr_out)
{
    const orc_a_params* p = &s->p;
    const int C = p->C, P = p->P, F = p->F, nv = C + F, j = s->st.j, fc = s->st.fc;
    const double dt = p->dt, eta = s->eta;
    const double pos = axis == 0 ? s->st.x : s->st.y, vel = axis == 0 ? s->st.xd : s->st.yd, zmp = axis == 0 ? s->st.xz : s->st.yz;
    const double cur = axis == 0 ? s->st.cur_x : s->st.cur_y;
    const double* fs = axis == 0 ? s->fsx : s->fsy;
    const double* cl = axis == 0 ? s->clx : s->cly;
    const int nC = 1 + C + F;
    double* H = (double*)calloc((size_t)nv * nv, 8);
    double* g = (double*)calloc(nv, 8);
    double* A = (double*)calloc((size_t)nC * nv, 8);
    double* lb = (double*)malloc(8 * nC); double* ub = (double*)malloc(8 * nC);

    /* stability row, :227-242 (xfs_store(fsCounter) == current footstep) */
    const double lambda = exp(-eta * dt);
    double anticip = 0.0;
    for (int i = C + 1; i <= P; ++i) anticip += exp(-eta * dt * i) * (1 - exp(-eta * dt)) * (cl[j + i] - cur);
    anticip += exp(-eta * dt * P) * (cl[P] - cur);
    for (int i = 0; i < C; ++i)
        A[i] = (1 / eta) * (1 - lambda) / (1 - pow(lambda, C)) * exp(-eta * dt * i) - dt * 1.0 * exp(-eta * dt * C);
    lb[0] = ub[0] = pos + vel / eta - zmp - anticip;

    /* ZMP rows, :173-181 : Pzmp u - mapping(:,2:end) f  in  [-z - w/2 + m1 cur, -z + w/2 + m1 cur] */
    for (int i = 1; i <= C; ++i) {
        double* row = &A[(size_t)i * nv];
        for (int k = 0; k < i; ++k) row[k] = dt;
        for (int k = 1; k <= F; ++k) row[C + k - 1] = -mapping[(i-1)*(F+1) + k];
        const double m1 = mapping[(i-1)*(F+1) + 0];
        ub[i] = 1.0 * (-zmp + p->w / 2) + m1 * cur;
        lb[i] = -(-1.0 * (-zmp - p->w / 2) - m1 * cur);
    }
    /* kinematic rows, :187-222 : difference_matrix f in [-b_lo, b_up] */
    for (int r = 1; r <= F; ++r) {
        double* row = &A[(size_t)(C + r) * nv];
        row[C + r - 1] = 1.0;
        if (r >= 2) row[C + r - 2] = -1.0;
        double bup = axis == 0 ? p->disp_forw : (p->disp_L / 2 + p->disp_L / 2);
        double blo = bup;
        if (fc == 1 && r == 1) {
            bup = axis == 0 ? p->disp_forw_dummy : (p->disp_L / 2 + p->disp_L / 2);
            blo = bup;
        }
        if (r == 1) { bup = bup + cur; blo = blo - cur; }
        ub[C + r] = bup; lb[C + r] = -blo;
    }
    /* cost, :268-276 */
    for (int i = 0; i < C; ++i) H[(size_t)i * nv + i] = 1.0;
    for (int k = 0; k < F; ++k) { H[(size_t)(C + k) * nv + C + k] = p->Qf; g[C + k] = -p->Qf * fs[fc + 1 + k]; }

    int nwsr = 2000;
    int rv = s->qp(nv, nC, H, g, A, lb, ub, sol, &nwsr);
    *nwsr_out = nwsr;
    free(H); free(g); free(A); free(lb); free(ub);
    return rv;
}

/* Structured view of the same per-axis QP (diagnostics / prototyping): stability row a, rhs b, ZMP band
 * [zlo, zhi] per sample, mapping (C x (F+1)), kinematic bounds and the footstep reference. */
int orc_a_axis_data(orc_a_sim* s, int axis, double* a, double* b, double* zlo, double* zhi, double* mapping_out,
                    double* klo, double* khi, double* pref)
{
    const orc_a_params* p = &s->p;
    const int C = p->C, P = p->P, F = p->F, j = s->st.j, fc = s->st.fc, ds = p->ds;
    const double dt = p->dt, eta = s->eta;
    const double pos = axis == 0 ? s->st.x : s->st.y, vel = axis == 0 ? s->st.xd : s->st.yd, zmp = axis == 0 ? s->st.xz : s->st.yz;
    const double cur = axis == 0 ? s->st.cur_x : s->st.cur_y;
    const double* fs = axis == 0 ? s->fsx : s->fsy;
    const double* cl = axis == 0 ? s->clx : s->cly;
    int pf = 0;
    memset(mapping_out, 0, sizeof(double) * C * (F + 1));
    for (int i = 1; i <= C; ++i) {
        if (j + i >= fs_timing(s, fc + pf + 1)) pf = pf + 1;
        const int rem = fs_timing(s, fc + pf + 1) - (j + i);
        if (pf + 1 > F + 1 || (rem <= ds && pf + 2 > F + 1)) return -2;
        if (rem > ds) mapping_out[(i-1)*(F+1) + pf] = 1;
        else { mapping_out[(i-1)*(F+1) + pf] = (double)rem / ds; mapping_out[(i-1)*(F+1) + pf + 1] = 1 - (double)rem / ds; }
    }
    const double lambda = exp(-eta * dt);
    double anticip = 0.0;
    for (int i = C + 1; i <= P; ++i) anticip += exp(-eta * dt * i) * (1 - exp(-eta * dt)) * (cl[j + i] - cur);
    anticip += exp(-eta * dt * P) * (cl[P] - cur);
    for (int i = 0; i < C; ++i)
        a[i] = (1 / eta) * (1 - lambda) / (1 - pow(lambda, C)) * exp(-eta * dt * i) - dt * 1.0 * exp(-eta * dt * C);
    *b = pos + vel / eta - zmp - anticip;
    for (int i = 1; i <= C; ++i) {
        const double m1 = mapping_out[(i-1)*(F+1)];
        zhi[i-1] = 1.0 * (-zmp + p->w / 2) + m1 * cur;
        zlo[i-1] = -(-1.0 * (-zmp - p->w / 2) - m1 * cur);
    }
    for (int r = 1; r <= F; ++r) {
        double bup = axis == 0 ? p->disp_forw : (p->disp_L / 2 + p->disp_L / 2), blo;
        if (fc == 1 && r == 1) bup = axis == 0 ? p->disp_forw_dummy : (p->disp_L / 2 + p->disp_L / 2);
        blo = bup;
        if (r == 1) { bup = bup + cur; blo = blo - cur; }
        khi[r-1] = bup; klo[r-1] = -blo; pref[r-1] = fs[fc + r];
    }
    return 0;
}


/* ---- swing-foot re-placement: trotting/quad_as_bip_no_plots.m:332-426 + compute_two_feet1.m,
 *                               walking/quad_walk_no_plots.m:336-504 + compute_one_feet_walk.m:84-140 ---- */
void orc_a_enable_feet(orc_a_sim* s, const orc_a_gait* g, const double* foot_plan /* rows x 8, 0-based rows = MATLAB row-1 */, int rows)
{
    free(s->fp);
    s->fprows = rows;
    s->fp = (double*)calloc((size_t)(rows + 12) * 8, 8);
    memcpy(s->fp + 8, foot_plan, sizeof(double) * (size_t)rows * 8);
    for (int r = rows + 1; r < rows + 12; ++r) memcpy(s->fp + (size_t)r * 8, foot_plan + (size_t)(rows - 1) * 8, 64);   /* MATLAB would auto-grow; never read */
    s->gait = g->gait; s->phi = g->phi; s->disp_i = g->disp_i; s->disp_o = g->disp_o; s->disp_forw_feet = g->disp_forw;
    s->counter = 1;
}
int orc_a_foot_rows(const orc_a_sim* s) { return s->fprows; }
void orc_a_get_foot_plan(const orc_a_sim* s, double* dst) { memcpy(dst, s->fp + 8, sizeof(double) * (size_t)s->fprows * 8); }

#define FPL(r, c) s->fp[(size_t)(r) * 8 + ((c) - 1)]
/* line through the two fixed feet, the line of opposite slope through the predicted footstep centre ("zmp"),
 * their intersection; returns slope m of the fixed diagonal, (dist_x, dist_y) = zmp - intersection */
static void fixed_diagonal(double fx1, double fy1, double fx2, double fy2, double zx, double zy, double* m, double* dx, double* dy)
{
    const double mm = (fy2 - fy1) / (fx2 - fx1), q = fy1 - mm * fx1;           /* polyfit(...,1) on two points */
    const double xi = (zy + mm * zx - q) / (2 * mm), yi = mm * xi + q;
    *m = mm; *dx = zx - xi; *dy = zy - yi;
}
static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void foot_tick_trot(orc_a_sim* s, int fc, double zx, double zy)
{
    const double di = s->disp_i, dobd = s->disp_o, df = s->disp_forw_feet;
    const int odd = (fc % 2) == 1;
    /* moving pair at row fc+1: odd -> BL(1,2), FR(5,6) ; even -> BR(3,4), FL(7,8).  fixed pair at row fc: the other two */
    const int f1 = odd ? 3 : 1, f2 = odd ? 7 : 5, m1 = odd ? 1 : 3, m2 = odd ? 5 : 7;
    double m, dx, dy;
    fixed_diagonal(FPL(fc, f1), FPL(fc, f1 + 1), FPL(fc, f2), FPL(fc, f2 + 1), zx, zy, &m, &dx, &dy);
    double x1, y1, x2, y2;
    const double fr1x = FPL(fc + 1, m1), fr1y = FPL(fc + 1, m1 + 1), fr2x = FPL(fc + 1, m2), fr2y = FPL(fc + 1, m2 + 1);
    if (s->phi == 3.14159265358979323846 / 2) {
        x1 = fr1x; x2 = fr2x;
        y1 = zy - m * (x1 - zx); y2 = zy - m * (x2 - zx);
    } else {
        const double tp = tan(s->phi);
        x1 = (zy + m * zx - fr1y + tp * fr1x) / (tp + m); y1 = tp * (x1 - fr1x) + fr1y;
        x2 = (zy + m * zx - fr2y + tp * fr2x) / (tp + m); y2 = tp * (x2 - fr2x) + fr2y;
    }
    if (dy != 0 || dx != 0) {                                                  /* changed: foot_plan(fc+1,:) = quattro_piedi */
        FPL(fc + 1, m1) = x1; FPL(fc + 1, m1 + 1) = y1; FPL(fc + 1, m2) = x2; FPL(fc + 1, m2 + 1) = y2;
        FPL(fc + 1, f1) = FPL(fc, f1); FPL(fc + 1, f1 + 1) = FPL(fc, f1 + 1); FPL(fc + 1, f2) = FPL(fc, f2); FPL(fc + 1, f2 + 1) = FPL(fc, f2 + 1);
    }
    /* quadprog(eye(4), -target, A, b): separable, so the minimiser is the projection on the box */
    const double lim_o = (fc == 1) ? dobd / 2 : dobd, lim_i = (fc == 1) ? di / 2 : di, lim_f = (fc == 1) ? df / 2 : df;
    /* first foot of the pair: y in [y_prev - disp_i, y_prev + disp_o], x <= x_prev + disp_forw ; second: y in [y_prev - disp_o, y_prev + disp_i] */
    {
        const double px = FPL(fc, m1), py = FPL(fc, m1 + 1);
        FPL(fc + 1, m1 + 1) = clipd(FPL(fc + 1, m1 + 1), py - lim_i, py + lim_o);
        if (FPL(fc + 1, m1) > px + lim_f) FPL(fc + 1, m1) = px + lim_f;
    }
    {
        const double px = FPL(fc, m2), py = FPL(fc, m2 + 1);
        FPL(fc + 1, m2 + 1) = clipd(FPL(fc + 1, m2 + 1), py - lim_o, py + lim_i);
        if (FPL(fc + 1, m2) > px + lim_f) FPL(fc + 1, m2) = px + lim_f;
    }
}

static void foot_tick_walk(orc_a_sim* s, int fc, double zx, double zy)
{
    const int counter = s->counter;
    if (!(counter == 2 || counter == 4 || counter == 6 || counter == 8)) return;
    const double di = s->disp_i, dobd = s->disp_o, df = s->disp_forw_feet;
    /* moving foot column, the two "fixed" feet whose diagonal is used (first two pairs of the `fixed` argument) */
    int mc, a1, a2; int outer_up;                       /* outer_up: upper y bound uses disp_o (left feet) else disp_i */
    if (counter == 2)      { mc = 7; a1 = 1; a2 = 5; outer_up = 1; }        /* FL moves; fixed = BL, FR, BR */
    else if (counter == 4) { mc = 3; a1 = 1; a2 = 5; outer_up = 0; }        /* BR moves; fixed = BL, FR, FL */
    else if (counter == 6) { mc = 5; a1 = 3; a2 = 7; outer_up = 0; }        /* FR moves; fixed = BR, FL, BL */
    else                   { mc = 1; a1 = 3; a2 = 7; outer_up = 1; }        /* BL moves; fixed = BR, FL, FR */
    double m, dx, dy;
    fixed_diagonal(FPL(fc, a1), FPL(fc, a1 + 1), FPL(fc, a2), FPL(fc, a2 + 1), zx, zy, &m, &dx, &dy);
    const double xfree = FPL(fc + 1, mc) + dx, yfree = FPL(fc + 1, mc + 1) + dy;
    if (dy != 0 || dx != 0)
        for (int l = 1; l <= 8; ++l) { FPL(fc + l, mc) = xfree; FPL(fc + l, mc + 1) = yfree; }
    const int dummy = (counter == 2 || counter == 4) && fc <= 4;
    const double lo_ = dummy ? dobd / 2 : dobd, li_ = dummy ? di / 2 : di, lf_ = dummy ? df / 2 : df;
    const double px = FPL(fc, mc), py = FPL(fc, mc + 1);
    double X1 = FPL(fc + 1, mc), X2 = FPL(fc + 1, mc + 1);
    if (outer_up) X2 = clipd(X2, py - li_, py + lo_); else X2 = clipd(X2, py - lo_, py + li_);
    if (X1 > px + lf_) X1 = px + lf_;
    if (counter == 8) {                                  /* quad_walk_no_plots.m:498-503: the y write-back only touches row fc+1 */
        for (int l = 1; l <= 8; ++l) FPL(fc + l, mc) = X1;
        FPL(fc + 1, mc + 1) = X2;
    } else {
        for (int l = 1; l <= 8; ++l) { FPL(fc + l, mc) = X1; FPL(fc + l, mc + 1) = X2; }
    }
}
#undef FPL

/* foot_*.txt rows (quad_as_bip_no_plots.m:482-509 / quad_walk_no_plots.m:562-613): dst = 4 x nrows x 3, order fl, fr, rl, rr */
int orc_a_foot_trajectories(const orc_a_sim* s, int sim_duration, double* dst)
{
    const int step = s->p.step, nsteps = sim_duration / step, nrows = nsteps * step;
    int row = 0, cont = 1;
#define FPL(r, c) s->fp[(size_t)(r) * 8 + ((c) - 1)]
#define PUT(foot, X, Y, Z) do { double* d_ = dst + ((size_t)(foot) * nrows + row) * 3; d_[0] = (X); d_[1] = (Y); d_[2] = (Z); } while (0)
    for (int i = 1; i <= nsteps; ++i) {
        if (s->gait == 0) {
            const int hold = step - 50;                                           /* the script hard-codes 30 + 50 */
            for (int k = 1; k <= hold; ++k) {
                PUT(0, FPL(i,7), FPL(i,8), 0.0); PUT(3, FPL(i,3), FPL(i,4), 0.0); PUT(1, FPL(i,5), FPL(i,6), 0.0); PUT(2, FPL(i,1), FPL(i,2), 0.0); ++row;
            }
            for (int j = 1; j <= 50; ++j) {
                const double z = -0.000032 * j * j + 0.0016 * j;
                if (i % 2 == 1) {
                    PUT(0, FPL(i,7), FPL(i,8), 0.0); PUT(3, FPL(i,3), FPL(i,4), 0.0);
                    PUT(2, FPL(i,1) + (FPL(i+1,1) - FPL(i,1)) / 50 * j, FPL(i,2) + (FPL(i+1,2) - FPL(i,2)) / 50 * j, z);
                    PUT(1, FPL(i,5) + (FPL(i+1,5) - FPL(i,5)) / 50 * j, FPL(i,6) + (FPL(i+1,6) - FPL(i,6)) / 50 * j, z);
                } else {
                    PUT(2, FPL(i,1), FPL(i,2), 0.0); PUT(1, FPL(i,5), FPL(i,6), 0.0);
                    PUT(0, FPL(i,7) + (FPL(i+1,7) - FPL(i,7)) / 50 * j, FPL(i,8) + (FPL(i+1,8) - FPL(i,8)) / 50 * j, z);
                    PUT(3, FPL(i,3) + (FPL(i+1,3) - FPL(i,3)) / 50 * j, FPL(i,4) + (FPL(i+1,4) - FPL(i,4)) / 50 * j, z);
                }
                ++row;
            }
        } else {
            for (int k = 1; k <= step; ++k) {
                const double z = -0.000032 * k * k + 0.0016 * k;
                const int mv = (cont == 2) ? 7 : (cont == 4) ? 3 : (cont == 6) ? 5 : (cont == 8) ? 1 : 0;
                const int cols[4] = {7, 5, 1, 3};                                 /* fl, fr, rl, rr */
                for (int ft = 0; ft < 4; ++ft) {
                    const int cc = cols[ft];
                    if (cc == mv) PUT(ft, FPL(i,cc) + (FPL(i+1,cc) - FPL(i,cc)) / step * k, FPL(i,cc+1) + (FPL(i+1,cc+1) - FPL(i,cc+1)) / step * k, z);
                    else PUT(ft, FPL(i,cc), FPL(i,cc+1), 0.0);
                }
                ++row;
            }
            cont = (cont == 8) ? 1 : cont + 1;
        }
    }
#undef PUT
#undef FPL
    return nrows;
}

/* One iteration of `for j = 1:sim_duration` (push = impulsive velocity disturbance added first, :134-148). */
int orc_a_tick(orc_a_sim* s, double push_x, double push_y, orc_a_tick_out* out, double* sol_x, double* sol_y)
{
    const orc_a_params* p = &s->p;
    const int C = p->C, F = p->F, j = s->st.j, fc = s->st.fc, ds = p->ds;
    memset(out, 0, sizeof(*out));
    out->fc = fc;
    out->com_before[0] = s->st.x; out->com_before[1] = s->st.y;        /* x_store(j), y_store(j) */
    s->st.xd += push_x; s->st.yd += push_y;

    /* mapping, :153-171 */
    double* mapping = (double*)calloc((size_t)C * (F + 2), 8);       /* one spare column: the .m matrix auto-grows */
    int overflow = 0;
    {
        double* m = (double*)calloc((size_t)C * (F + 1), 8);
        int pf = 0;
        for (int i = 1; i <= C; ++i) {
            if (j + i >= fs_timing(s, fc + pf + 1)) pf = pf + 1;
            const int rem = fs_timing(s, fc + pf + 1) - (j + i);
            if (pf + 1 > F + 1 || (rem <= ds && pf + 2 > F + 1)) { overflow = 1; break; }
            if (rem > ds) m[(i-1)*(F+1) + pf] = 1;
            else { m[(i-1)*(F+1) + pf] = (double)rem / ds; m[(i-1)*(F+1) + pf + 1] = 1 - (double)rem / ds; }
        }
        free(mapping); mapping = m;
    }
    if (overflow) { free(mapping); out->rv[0] = out->rv[1] = -2; return -2; }   /* horizon spans more than F-1 boundaries */

    double* sx = sol_x ? sol_x : (double*)malloc(8 * (C + F));
    double* sy = sol_y ? sol_y : (double*)malloc(8 * (C + F));
    out->rv[0] = solve_axis(s, 0, mapping, sx, &out->nwsr[0]);
    out->rv[1] = solve_axis(s, 1, mapping, sy, &out->nwsr[1]);
    free(mapping);
    const double ux = sx[0], uy = sy[0];
    s->st.pred_x = sx[C]; s->st.pred_y = sy[C];                       /* predicted_xfs(1), predicted_yfs(1) */
    out->u0[0] = ux; out->u0[1] = uy; out->f0[0] = sx[C]; out->f0[1] = sy[C];
    if (!sol_x) free(sx);
    if (!sol_y) free(sy);

    /* state update, :297-322 */
    const double* Au = s->A_upd; const double* Bu = s->B_upd;
    {
        const double a = s->st.x, b = s->st.xd, c = s->st.xz;
        s->st.x  = (Au[0]*a + Au[1]*b + Au[2]*c) + Bu[0]*ux;
        s->st.xd = (Au[3]*a + Au[4]*b + Au[5]*c) + Bu[1]*ux;
        s->st.xz = (Au[6]*a + Au[7]*b + Au[8]*c) + Bu[2]*ux;
    }
    {
        const double a = s->st.y, b = s->st.yd, c = s->st.yz;
        s->st.y  = (Au[0]*a + Au[1]*b + Au[2]*c) + Bu[0]*uy;
        s->st.yd = (Au[3]*a + Au[4]*b + Au[5]*c) + Bu[1]*uy;
        s->st.yz = (Au[6]*a + Au[7]*b + Au[8]*c) + Bu[2]*uy;
    }
    out->vel_after[0] = s->st.xd; out->vel_after[1] = s->st.yd;       /* xd_store(j), yd_store(j) */

    /* second quadprog: swing feet of the NEXT footstep row (uses this tick's predicted footstep and the old fsCounter) */
    if (s->fp && fc + 9 <= s->fprows + 10) {
        if (s->gait == 0) foot_tick_trot(s, fc, s->st.pred_x, s->st.pred_y);
        else foot_tick_walk(s, fc, s->st.pred_x, s->st.pred_y);
    }
    /* footstep bookkeeping, :522-556 */
    if (j + 1 >= fs_timing(s, fc + 1)) {
        const int nfc = fc + 1;
        s->counter += 1;                                                 /* quad_walk_no_plots.m:527 */
        s->st.fc = nfc;
        s->st.cur_x = s->st.pred_x; s->st.cur_y = s->st.pred_y;
        const double dx = s->st.pred_x - s->fsx[nfc], dy = s->st.pred_y - s->fsy[nfc];
        for (int i = 1; i <= s->nplan; ++i) { s->fsx[i] = s->fsx[i] + dx; s->fsy[i] = s->fsy[i] + dy; }
        build_centerline(s, 0);
        out->stepped = 1;
    }
    s->st.j = j + 1;
    return (out->rv[0] != 0 || out->rv[1] != 0) ? 1 : 0;
}

int orc_a_run(orc_a_sim* s, int ticks, orc_a_tick_out* outs)
{
    int bad = 0;
    for (int t = 0; t < ticks; ++t) bad += orc_a_tick(s, 0.0, 0.0, &outs[t], NULL, NULL) != 0;
    return bad;
}
