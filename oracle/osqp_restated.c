/*
 * oracle/osqp_restated.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU (FP64, single thread) restatement of the OSQP algorithm that the reference
 * delegates its per-time-step QP to:
 *     /root/reference/src/fnc/controller/PredictiveControllers.py:259-283
 *     (MPC.osqp_solve_qp: OSQP().setup(P,q,A,l,u,verbose=False,polish=True); solve())
 *
 * `osqp` is a third-party PyPI package that is NOT vendored under /root/reference and is
 * NOT installed in this image (version unpinned: reference README.md:18 "pip install osqp").
 * This file restates its PUBLISHED algorithm (Stellato, Banjac, Goulart, Bemporad, Boyd,
 * "OSQP: an operator splitting solver for quadratic programs", Math. Prog. Comp. 2020, and
 * the documented default settings) from memory:
 *     - modified Ruiz equilibration (scaling = 10) with cost scaling,
 *     - ADMM iteration with relaxation alpha = 1.6, sigma = 1e-6, rho = 0.1,
 *       rho_eq = 1e3 * rho on equality rows,
 *     - quasi-definite KKT system solved by a sparse LDL' factorisation
 *       (up-looking, elimination-tree based -- the algorithm of T. Davis' LDL / QDLDL),
 *     - termination on unscaled residuals every 25 iterations (eps_abs = eps_rel = 1e-3),
 *     - adaptive rho (tolerance 5) every `adaptive_rho_interval` iterations
 *       (fixed 50: the OSQP >= 1.0 default; 0.6.x picks it from wall-clock timing and is
 *       therefore not reproducible),
 *     - primal / dual infeasibility certificates,
 *     - polish (delta = 1e-6, 3 refinement iterations), accepted only if it improves
 *       the residuals.
 * PARITY WITH THE REAL `osqp` BINARY IS UNPINNED (no osqp here, no golden vectors upstream).
 * What pins the result instead is solver independent: tests/ check a KKT optimality
 * certificate on every returned (x, y) -- see oracle/lmpc_oracle.py:kkt_certificate.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build:  gcc -O2 -fPIC -shared -o oracle/libosqp_restated.so oracle/osqp_restated.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define OQ_INF 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_TOL 1e-4
#define RHO_EQ_OVER_RHO_INEQ 1e3

typedef struct {
    double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, delta;
    double adaptive_rho_tolerance;
    int max_iter, check_termination, scaling, adaptive_rho, adaptive_rho_interval;
    int polish, polish_refine_iter;
} oq_settings;

typedef struct {
    int iter, status, status_polish, rho_updates;
    double obj_val, pri_res, dua_res, rho_estimate;
} oq_info;

void oq_default_settings(oq_settings *s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
    s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
    s->delta = 1e-6; s->adaptive_rho_tolerance = 5.0;
    s->max_iter = 4000; s->check_termination = 25; s->scaling = 10;
    s->adaptive_rho = 1; s->adaptive_rho_interval = 50;
    s->polish = 0; s->polish_refine_iter = 3;
}

/* ------------------------------------------------------------------ sparse helpers */
typedef struct { int n, nnz; int *p, *i; double *x; } csc;   /* n columns */

static csc *csc_alloc(int ncol, int nnz) {
    csc *M = (csc *)malloc(sizeof(csc));
    M->n = ncol; M->nnz = nnz;
    M->p = (int *)calloc(ncol + 1, sizeof(int));
    M->i = (int *)malloc(sizeof(int) * (nnz > 0 ? nnz : 1));
    M->x = (double *)malloc(sizeof(double) * (nnz > 0 ? nnz : 1));
    return M;
}
static void csc_free(csc *M) { if (M) { free(M->p); free(M->i); free(M->x); free(M); } }

/* triplets (r,c,v) -> CSC with sorted rows, duplicates summed */
static csc *triplet_to_csc(int ncol, int nt, const int *r, const int *c, const double *v) {
    csc *M = csc_alloc(ncol, nt);
    int *cnt = (int *)calloc(ncol + 1, sizeof(int));
    for (int k = 0; k < nt; k++) cnt[c[k] + 1]++;
    for (int j = 0; j < ncol; j++) cnt[j + 1] += cnt[j];
    memcpy(M->p, cnt, sizeof(int) * (ncol + 1));
    int *pos = (int *)malloc(sizeof(int) * (ncol + 1));
    memcpy(pos, cnt, sizeof(int) * (ncol + 1));
    for (int k = 0; k < nt; k++) { int q = pos[c[k]]++; M->i[q] = r[k]; M->x[q] = v[k]; }
    /* insertion sort inside each column, then merge duplicates */
    int w = 0; int *np_ = (int *)calloc(ncol + 1, sizeof(int));
    for (int j = 0; j < ncol; j++) {
        int a = M->p[j], b = M->p[j + 1];
        for (int k = a + 1; k < b; k++) {
            int ri = M->i[k]; double xv = M->x[k]; int q = k - 1;
            while (q >= a && M->i[q] > ri) { M->i[q + 1] = M->i[q]; M->x[q + 1] = M->x[q]; q--; }
            M->i[q + 1] = ri; M->x[q + 1] = xv;
        }
        np_[j] = w;
        for (int k = a; k < b; k++) {
            if (w > np_[j] && M->i[w - 1] == M->i[k]) M->x[w - 1] += M->x[k];
            else { M->i[w] = M->i[k]; M->x[w] = M->x[k]; w++; }
        }
    }
    np_[ncol] = w; memcpy(M->p, np_, sizeof(int) * (ncol + 1)); M->nnz = w;
    free(cnt); free(pos); free(np_);
    return M;
}

/* ------------------------------------------------------------------ LDL' (quasi-definite) */
typedef struct {
    int n; int *Lp, *Li, *Parent, *Lnz, *Flag, *Pattern; double *Lx, *D, *Y;
} ldl;

static void ldl_free(ldl *F) {
    if (!F) return;
    free(F->Lp); free(F->Li); free(F->Parent); free(F->Lnz); free(F->Flag); free(F->Pattern);
    free(F->Lx); free(F->D); free(F->Y); free(F);
}

/* K: upper triangle (row <= col), CSC, sorted. Symbolic: elimination tree + column counts. */
static ldl *ldl_symbolic(const csc *K) {
    int n = K->n;
    ldl *F = (ldl *)calloc(1, sizeof(ldl));
    F->n = n;
    F->Lp = (int *)calloc(n + 1, sizeof(int)); F->Parent = (int *)malloc(sizeof(int) * n);
    F->Lnz = (int *)calloc(n, sizeof(int)); F->Flag = (int *)malloc(sizeof(int) * n);
    F->Pattern = (int *)malloc(sizeof(int) * n);
    F->D = (double *)malloc(sizeof(double) * n); F->Y = (double *)calloc(n, sizeof(double));
    for (int k = 0; k < n; k++) {
        F->Parent[k] = -1; F->Flag[k] = k; F->Lnz[k] = 0;
        for (int p = K->p[k]; p < K->p[k + 1]; p++) {
            int i = K->i[p];
            if (i < k) {
                for (; F->Flag[i] != k; i = F->Parent[i]) {
                    if (F->Parent[i] == -1) F->Parent[i] = k;
                    F->Lnz[i]++; F->Flag[i] = k;
                }
            }
        }
    }
    for (int k = 0; k < n; k++) F->Lp[k + 1] = F->Lp[k] + F->Lnz[k];
    int lnz = F->Lp[n];
    F->Li = (int *)malloc(sizeof(int) * (lnz > 0 ? lnz : 1));
    F->Lx = (double *)malloc(sizeof(double) * (lnz > 0 ? lnz : 1));
    return F;
}

/* numeric up-looking LDL'; returns 0 on success, -1 on zero pivot */
static int ldl_numeric(ldl *F, const csc *K) {
    int n = F->n;
    for (int k = 0; k < n; k++) {
        F->Y[k] = 0.0; int top = n; F->Flag[k] = k; F->Lnz[k] = 0;
        for (int p = K->p[k]; p < K->p[k + 1]; p++) {
            int i = K->i[p];
            if (i <= k) {
                F->Y[i] += K->x[p];
                int len = 0;
                for (; F->Flag[i] != k; i = F->Parent[i]) { F->Pattern[len++] = i; F->Flag[i] = k; }
                while (len > 0) F->Pattern[--top] = F->Pattern[--len];
            }
        }
        F->D[k] = F->Y[k]; F->Y[k] = 0.0;
        for (; top < n; top++) {
            int i = F->Pattern[top]; double yi = F->Y[i]; F->Y[i] = 0.0;
            int p2 = F->Lp[i] + F->Lnz[i], p;
            for (p = F->Lp[i]; p < p2; p++) F->Y[F->Li[p]] -= F->Lx[p] * yi;
            double lki = yi / F->D[i];
            F->D[k] -= lki * yi;
            F->Li[p] = k; F->Lx[p] = lki; F->Lnz[i]++;
        }
        if (F->D[k] == 0.0) return -1;
    }
    return 0;
}

static void ldl_solve(const ldl *F, double *b) {
    int n = F->n;
    for (int j = 0; j < n; j++) { double bj = b[j]; for (int p = F->Lp[j]; p < F->Lp[j + 1]; p++) b[F->Li[p]] -= F->Lx[p] * bj; }
    for (int j = 0; j < n; j++) b[j] /= F->D[j];
    for (int j = n - 1; j >= 0; j--) { double s = b[j]; for (int p = F->Lp[j]; p < F->Lp[j + 1]; p++) s -= F->Lx[p] * b[F->Li[p]]; b[j] = s; }
}

/* ------------------------------------------------------------------ KKT system object */
typedef struct {
    int n, m, N;             /* N = n + mrows actually used */
    csc *K;                  /* permuted upper triangle */
    int *perm, *iperm;       /* perm[new] = old ; iperm[old] = new */
    int *diag_pos;           /* position in K->x of the (n+i, n+i) diagonal entry for row i */
    ldl *F; double *work;
} kkt_t;

static void kkt_free(kkt_t *S) {
    if (!S) return; csc_free(S->K); free(S->perm); free(S->iperm); free(S->diag_pos); ldl_free(S->F); free(S->work); free(S);
}

/* Build [[P + sig I, A_sel'],[A_sel, diag(dvals)]] for selected rows `rows` (nr of them) of A.
 * P: upper triangle CSC. A: CSC (m x n). perm_full: permutation over n+m (perm[new]=old) or NULL. */
static kkt_t *kkt_build(int n, int m, const int *Pp, const int *Pi, const double *Px,
                        const int *Ap, const int *Ai, const double *Ax, double sig,
                        int nr, const int *rows, const double *dvals, const int *perm_full) {
    kkt_t *S = (kkt_t *)calloc(1, sizeof(kkt_t));
    int N = n + nr; S->n = n; S->m = nr; S->N = N;
    int *rowmap = (int *)malloc(sizeof(int) * (m > 0 ? m : 1));
    for (int i = 0; i < m; i++) rowmap[i] = -1;
    for (int k = 0; k < nr; k++) rowmap[rows[k]] = k;
    /* permutation restricted to the kept indices, relative order preserved */
    S->perm = (int *)malloc(sizeof(int) * N); S->iperm = (int *)malloc(sizeof(int) * N);
    if (perm_full) {
        int w = 0;
        for (int k = 0; k < n + m; k++) {
            int o = perm_full[k];
            if (o < n) S->perm[w++] = o;
            else if (rowmap[o - n] >= 0) S->perm[w++] = n + rowmap[o - n];
        }
    } else for (int k = 0; k < N; k++) S->perm[k] = k;
    for (int k = 0; k < N; k++) S->iperm[S->perm[k]] = k;

    int nt = Pp[n] + n + Ap[n] + nr;
    int *tr = (int *)malloc(sizeof(int) * nt), *tc = (int *)malloc(sizeof(int) * nt);
    double *tv = (double *)malloc(sizeof(double) * nt);
    int t = 0;
#define PUSH(r_, c_, v_) do { int a_ = S->iperm[r_], b_ = S->iperm[c_]; if (a_ > b_) { int s_ = a_; a_ = b_; b_ = s_; } tr[t] = a_; tc[t] = b_; tv[t] = (v_); t++; } while (0)
    for (int j = 0; j < n; j++) for (int p = Pp[j]; p < Pp[j + 1]; p++) if (Pi[p] <= j) PUSH(Pi[p], j, Px[p]);
    for (int j = 0; j < n; j++) PUSH(j, j, sig);
    for (int j = 0; j < n; j++) for (int p = Ap[j]; p < Ap[j + 1]; p++) { int rr = rowmap[Ai[p]]; if (rr >= 0) PUSH(j, n + rr, Ax[p]); }
    int tdiag0 = t;
    for (int k = 0; k < nr; k++) PUSH(n + k, n + k, dvals[k]);
#undef PUSH
    (void)tdiag0;
    S->K = triplet_to_csc(N, t, tr, tc, tv);
    S->diag_pos = (int *)malloc(sizeof(int) * (nr > 0 ? nr : 1));
    for (int k = 0; k < nr; k++) {
        int c = S->iperm[n + k]; int pos = -1;
        for (int p = S->K->p[c]; p < S->K->p[c + 1]; p++) if (S->K->i[p] == c) pos = p;
        S->diag_pos[k] = pos;
    }
    S->F = ldl_symbolic(S->K);
    S->work = (double *)malloc(sizeof(double) * N);
    free(tr); free(tc); free(tv); free(rowmap);
    return S;
}
static int kkt_factor(kkt_t *S) { return ldl_numeric(S->F, S->K); }
static void kkt_set_diag(kkt_t *S, const double *dvals) { for (int k = 0; k < S->m; k++) S->K->x[S->diag_pos[k]] = dvals[k]; }
static void kkt_solve(kkt_t *S, double *b) {
    for (int k = 0; k < S->N; k++) S->work[k] = b[S->perm[k]];
    ldl_solve(S->F, S->work);
    for (int k = 0; k < S->N; k++) b[S->perm[k]] = S->work[k];
}

/* ------------------------------------------------------------------ vector helpers */
static double vinf(const double *v, int n) { double r = 0; for (int i = 0; i < n; i++) { double a = fabs(v[i]); if (a > r) r = a; } return r; }
static double vinf_scaled(const double *s, const double *v, int n) { double r = 0; for (int i = 0; i < n; i++) { double a = fabs(s[i] * v[i]); if (a > r) r = a; } return r; }
static void A_mul(int n, int m, const int *Ap, const int *Ai, const double *Ax, const double *x, double *y) {
    for (int i = 0; i < m; i++) y[i] = 0; for (int j = 0; j < n; j++) { double xj = x[j]; for (int p = Ap[j]; p < Ap[j + 1]; p++) y[Ai[p]] += Ax[p] * xj; }
}
static void At_mul(int n, const int *Ap, const int *Ai, const double *Ax, const double *y, double *x) {
    for (int j = 0; j < n; j++) { double s = 0; for (int p = Ap[j]; p < Ap[j + 1]; p++) s += Ax[p] * y[Ai[p]]; x[j] = s; }
}
/* y = P x with P given as upper triangle */
static void P_mul(int n, const int *Pp, const int *Pi, const double *Px, const double *x, double *y) {
    for (int i = 0; i < n; i++) y[i] = 0;
    for (int j = 0; j < n; j++) for (int p = Pp[j]; p < Pp[j + 1]; p++) {
        int i = Pi[p]; if (i > j) continue;
        y[i] += Px[p] * x[j]; if (i != j) y[j] += Px[p] * x[i];
    }
}
static double limit_scaling(double v) { if (v < MIN_SCALING) return 1.0; if (v > MAX_SCALING) return MAX_SCALING; return v; }

/* ------------------------------------------------------------------ the solver
 * P : n x n, CSC; only entries with row <= col are used (the reference passes the full
 *     symmetric matrix, PredictiveControllers.py:149,361 -- OSQP uses its upper triangle).
 * A : m x n CSC.  l,u : bounds, |.| >= 1e20 treated as infinite.
 * perm : fill-reducing permutation of the (n+m) KKT system (perm[new]=old) or NULL.
 * returns status: 1 solved, 2 solved inaccurate, -2 max iter, -3 primal infeasible,
 *                 -4 dual infeasible, -10 factorisation failure.                     */
int oq_solve(int n, int m, const int *Pp_in, const int *Pi_in, const double *Px_in, const double *q_in,
             const int *Ap, const int *Ai, const double *Ax_in, const double *l_in, const double *u_in,
             const int *perm, const oq_settings *st, double *x_out, double *y_out, double *z_out, oq_info *info) {
    int nnzP = Pp_in[n], nnzA = Ap[n];
    /* copy P keeping upper triangle only */
    int *Pp = (int *)calloc(n + 1, sizeof(int)); int *Pi = (int *)malloc(sizeof(int) * (nnzP + 1));
    double *Px = (double *)malloc(sizeof(double) * (nnzP + 1));
    { int w = 0; for (int j = 0; j < n; j++) { Pp[j] = w; for (int p = Pp_in[j]; p < Pp_in[j + 1]; p++) if (Pi_in[p] <= j) { Pi[w] = Pi_in[p]; Px[w] = Px_in[p]; w++; } } Pp[n] = w; nnzP = w; }
    double *Ax = (double *)malloc(sizeof(double) * (nnzA + 1)); memcpy(Ax, Ax_in, sizeof(double) * nnzA);
    double *q = (double *)malloc(sizeof(double) * n); memcpy(q, q_in, sizeof(double) * n);
    double *l = (double *)malloc(sizeof(double) * (m + 1)), *u = (double *)malloc(sizeof(double) * (m + 1));
    for (int i = 0; i < m; i++) { l[i] = l_in[i] < -1e20 ? -OQ_INF : l_in[i]; u[i] = u_in[i] > 1e20 ? OQ_INF : u_in[i]; }

    double *D = (double *)malloc(sizeof(double) * n), *E = (double *)malloc(sizeof(double) * (m + 1));
    double *Dinv = (double *)malloc(sizeof(double) * n), *Einv = (double *)malloc(sizeof(double) * (m + 1));
    for (int j = 0; j < n; j++) D[j] = 1.0; for (int i = 0; i < m; i++) E[i] = 1.0;
    double c = 1.0;
    double *dt = (double *)malloc(sizeof(double) * n), *et = (double *)malloc(sizeof(double) * (m + 1));

    /* ---- modified Ruiz equilibration ---- */
    for (int it = 0; it < st->scaling; it++) {
        for (int j = 0; j < n; j++) dt[j] = 0; for (int i = 0; i < m; i++) et[i] = 0;
        for (int j = 0; j < n; j++) for (int p = Pp[j]; p < Pp[j + 1]; p++) {   /* symmetric column norms */
            double a = fabs(Px[p]); int i = Pi[p];
            if (a > dt[j]) dt[j] = a; if (a > dt[i]) dt[i] = a;
        }
        for (int j = 0; j < n; j++) for (int p = Ap[j]; p < Ap[j + 1]; p++) {
            double a = fabs(Ax[p]); if (a > dt[j]) dt[j] = a; if (a > et[Ai[p]]) et[Ai[p]] = a;
        }
        for (int j = 0; j < n; j++) dt[j] = 1.0 / sqrt(limit_scaling(dt[j]));
        for (int i = 0; i < m; i++) et[i] = 1.0 / sqrt(limit_scaling(et[i]));
        for (int j = 0; j < n; j++) for (int p = Pp[j]; p < Pp[j + 1]; p++) Px[p] *= dt[j] * dt[Pi[p]];
        for (int j = 0; j < n; j++) for (int p = Ap[j]; p < Ap[j + 1]; p++) Ax[p] *= dt[j] * et[Ai[p]];
        for (int j = 0; j < n; j++) { q[j] *= dt[j]; D[j] *= dt[j]; }
        for (int i = 0; i < m; i++) E[i] *= et[i];
        /* cost scaling */
        for (int j = 0; j < n; j++) dt[j] = 0;
        for (int j = 0; j < n; j++) for (int p = Pp[j]; p < Pp[j + 1]; p++) { double a = fabs(Px[p]); int i = Pi[p]; if (a > dt[j]) dt[j] = a; if (a > dt[i]) dt[i] = a; }
        double mean = 0; for (int j = 0; j < n; j++) mean += dt[j]; mean /= (n > 0 ? n : 1);
        double ct = limit_scaling(mean), qn = limit_scaling(vinf(q, n));
        if (qn > ct) ct = qn; ct = 1.0 / ct;
        for (int p = 0; p < nnzP; p++) Px[p] *= ct; for (int j = 0; j < n; j++) q[j] *= ct;
        c *= ct;
    }
    for (int j = 0; j < n; j++) Dinv[j] = 1.0 / D[j];
    for (int i = 0; i < m; i++) { Einv[i] = 1.0 / E[i]; if (l[i] > -OQ_INF) l[i] *= E[i]; if (u[i] < OQ_INF) u[i] *= E[i]; }
    double cinv = 1.0 / c;

    /* ---- rho vector ---- */
    int *ctype = (int *)malloc(sizeof(int) * (m + 1));
    double *rho_vec = (double *)malloc(sizeof(double) * (m + 1)), *rho_inv = (double *)malloc(sizeof(double) * (m + 1));
    double *kd = (double *)malloc(sizeof(double) * (m + 1));
    double rho = st->rho; if (rho < RHO_MIN) rho = RHO_MIN; if (rho > RHO_MAX) rho = RHO_MAX;
    for (int i = 0; i < m; i++) {
        if (l[i] <= -OQ_INF && u[i] >= OQ_INF) ctype[i] = -1;
        else if (u[i] - l[i] < RHO_TOL) ctype[i] = 1; else ctype[i] = 0;
    }
#define SET_RHO() for (int i = 0; i < m; i++) { rho_vec[i] = ctype[i] == -1 ? RHO_MIN : (ctype[i] == 1 ? RHO_EQ_OVER_RHO_INEQ * rho : rho); rho_inv[i] = 1.0 / rho_vec[i]; kd[i] = -rho_inv[i]; }
    SET_RHO();
    int *allrows = (int *)malloc(sizeof(int) * (m + 1)); for (int i = 0; i < m; i++) allrows[i] = i;
    kkt_t *S = kkt_build(n, m, Pp, Pi, Px, Ap, Ai, Ax, st->sigma, m, allrows, kd, perm);
    int status = 0;
    if (kkt_factor(S) != 0) status = -10;

    double *x = (double *)calloc(n, sizeof(double)), *z = (double *)calloc(m + 1, sizeof(double)), *y = (double *)calloc(m + 1, sizeof(double));
    double *xp = (double *)calloc(n, sizeof(double)), *zp = (double *)calloc(m + 1, sizeof(double));
    double *rhs = (double *)calloc(n + m + 1, sizeof(double));
    double *dx = (double *)calloc(n, sizeof(double)), *dy = (double *)calloc(m + 1, sizeof(double));
    double *Axv = (double *)calloc(m + 1, sizeof(double)), *Pxv = (double *)calloc(n, sizeof(double)), *Aty = (double *)calloc(n, sizeof(double));
    double *tn = (double *)calloc(n, sizeof(double)), *tm = (double *)calloc(m + 1, sizeof(double));
    int iter = 0, rho_updates = 0; double pri_res = 0, dua_res = 0;
    const double alpha = st->alpha;

    for (iter = 1; status == 0 && iter <= st->max_iter; iter++) {
        memcpy(xp, x, sizeof(double) * n); memcpy(zp, z, sizeof(double) * m);
        for (int j = 0; j < n; j++) rhs[j] = st->sigma * xp[j] - q[j];
        for (int i = 0; i < m; i++) rhs[n + i] = zp[i] - rho_inv[i] * y[i];
        kkt_solve(S, rhs);
        for (int i = 0; i < m; i++) rhs[n + i] = zp[i] + rho_inv[i] * (rhs[n + i] - y[i]);   /* ztilde */
        for (int j = 0; j < n; j++) { x[j] = alpha * rhs[j] + (1 - alpha) * xp[j]; dx[j] = x[j] - xp[j]; }
        for (int i = 0; i < m; i++) {
            double zr = alpha * rhs[n + i] + (1 - alpha) * zp[i];
            double zn = zr + rho_inv[i] * y[i];
            if (zn < l[i]) zn = l[i]; if (zn > u[i]) zn = u[i];
            z[i] = zn; dy[i] = rho_vec[i] * (zr - zn); y[i] += dy[i];
        }
        int check = st->check_termination && (iter % st->check_termination == 0);
        int adapt = st->adaptive_rho && st->adaptive_rho_interval && (iter % st->adaptive_rho_interval == 0);
        if (!check && !adapt && iter != st->max_iter) continue;

        /* unscaled residuals */
        A_mul(n, m, Ap, Ai, Ax, x, Axv); P_mul(n, Pp, Pi, Px, x, Pxv); At_mul(n, Ap, Ai, Ax, y, Aty);
        for (int i = 0; i < m; i++) tm[i] = Axv[i] - z[i];
        pri_res = vinf_scaled(Einv, tm, m);
        for (int j = 0; j < n; j++) tn[j] = Pxv[j] + q[j] + Aty[j];
        dua_res = cinv * vinf_scaled(Dinv, tn, n);
        double nAx = vinf_scaled(Einv, Axv, m), nz = vinf_scaled(Einv, z, m);
        double nPx = cinv * vinf_scaled(Dinv, Pxv, n), nAty = cinv * vinf_scaled(Dinv, Aty, n), nq = cinv * vinf_scaled(Dinv, q, n);
        double pmax = nAx > nz ? nAx : nz; double dmax = nPx > nAty ? nPx : nAty; if (nq > dmax) dmax = nq;
        if (check || iter == st->max_iter) {
            double eps_p = st->eps_abs + st->eps_rel * pmax, eps_d = st->eps_abs + st->eps_rel * dmax;
            if (pri_res <= eps_p && dua_res <= eps_d) { status = 1; break; }
            /* primal infeasibility: dy certificate */
            /* project dy on the polar of the recession cone of [l,u], then test the certificate */
            for (int i = 0; i < m; i++) {
                double v = dy[i];
                if (u[i] >= OQ_INF) { if (l[i] <= -OQ_INF) v = 0; else if (v > 0) v = 0; }
                else if (l[i] <= -OQ_INF) { if (v < 0) v = 0; }
                tm[i] = v;
            }
            double ndy = vinf_scaled(E, tm, m);
            if (ndy > 1e-30) {
                double sup = 0;
                for (int i = 0; i < m; i++) { if (u[i] < OQ_INF && tm[i] > 0) sup += u[i] * tm[i]; if (l[i] > -OQ_INF && tm[i] < 0) sup += l[i] * tm[i]; }
                if (sup < -st->eps_prim_inf * ndy) {
                    At_mul(n, Ap, Ai, Ax, tm, tn);
                    if (vinf_scaled(Dinv, tn, n) < st->eps_prim_inf * ndy) { status = -3; break; }
                }
            }
            /* dual infeasibility: dx certificate */
            double ndx = vinf_scaled(D, dx, n);
            if (ndx > 1e-30) {
                double qdx = 0; for (int j = 0; j < n; j++) qdx += q[j] * dx[j];
                if (cinv * qdx < -st->eps_dual_inf * ndx) {
                    P_mul(n, Pp, Pi, Px, dx, tn);
                    if (cinv * vinf_scaled(Dinv, tn, n) < st->eps_dual_inf * ndx) {
                        A_mul(n, m, Ap, Ai, Ax, dx, tm); int ok = 1;
                        for (int i = 0; i < m && ok; i++) {
                            double a = Einv[i] * tm[i];
                            if ((u[i] < OQ_INF && a > st->eps_dual_inf * ndx) || (l[i] > -OQ_INF && a < -st->eps_dual_inf * ndx)) ok = 0;
                        }
                        if (ok) { status = -4; break; }
                    }
                }
            }
        }
        if (adapt && iter < st->max_iter) {
            double pn = pri_res / (pmax + 1e-10), dn = dua_res / (dmax + 1e-10);
            double rn = rho * sqrt(pn / (dn + 1e-10));
            if (rn < RHO_MIN) rn = RHO_MIN; if (rn > RHO_MAX) rn = RHO_MAX;
            if (rn > rho * st->adaptive_rho_tolerance || rn < rho / st->adaptive_rho_tolerance) {
                rho = rn; SET_RHO(); kkt_set_diag(S, kd);
                if (kkt_factor(S) != 0) { status = -10; break; }
                rho_updates++;
            }
        }
    }
    if (status == 0) {   /* ran out of iterations: "solved inaccurate" uses 10x tolerances */
        iter = st->max_iter;
        A_mul(n, m, Ap, Ai, Ax, x, Axv); P_mul(n, Pp, Pi, Px, x, Pxv); At_mul(n, Ap, Ai, Ax, y, Aty);
        double nAx = vinf_scaled(Einv, Axv, m), nz = vinf_scaled(Einv, z, m);
        double nPx = cinv * vinf_scaled(Dinv, Pxv, n), nAty = cinv * vinf_scaled(Dinv, Aty, n), nq = cinv * vinf_scaled(Dinv, q, n);
        double pmax = nAx > nz ? nAx : nz; double dmax = nPx > nAty ? nPx : nAty; if (nq > dmax) dmax = nq;
        status = (pri_res <= 10 * (st->eps_abs + st->eps_rel * pmax) && dua_res <= 10 * (st->eps_abs + st->eps_rel * dmax)) ? 2 : -2;
    }

    int status_polish = 0;
    /* ---- polish ---- */
    if (st->polish && status == 1) {
        int nlow = 0, nupp = 0; int *rows = (int *)malloc(sizeof(int) * (m + 1));
        for (int i = 0; i < m; i++) if (z[i] - l[i] < -y[i]) rows[nlow++] = i;
        for (int i = 0; i < m; i++) if (u[i] - z[i] < y[i]) { int dup = 0; if (l[i] == u[i]) for (int k = 0; k < nlow; k++) if (rows[k] == i) dup = 1; if (!dup) rows[nlow + nupp++] = i; }
        int nr = nlow + nupp;
        double *dv = (double *)malloc(sizeof(double) * (nr + 1)); for (int k = 0; k < nr; k++) dv[k] = -st->delta;
        kkt_t *R = kkt_build(n, m, Pp, Pi, Px, Ap, Ai, Ax, st->delta, nr, rows, dv, perm);
        if (kkt_factor(R) == 0) {
            int NR = n + nr;
            double *b = (double *)malloc(sizeof(double) * NR), *sol = (double *)malloc(sizeof(double) * NR), *r = (double *)malloc(sizeof(double) * NR);
            for (int j = 0; j < n; j++) b[j] = -q[j];
            for (int k = 0; k < nlow; k++) b[n + k] = l[rows[k]];
            for (int k = nlow; k < nr; k++) b[n + k] = u[rows[k]];
            int *rowmap = (int *)malloc(sizeof(int) * (m + 1)); for (int i = 0; i < m; i++) rowmap[i] = -1; for (int k = 0; k < nr; k++) rowmap[rows[k]] = k;
            memcpy(sol, b, sizeof(double) * NR); kkt_solve(R, sol);
            for (int itr = 0; itr < st->polish_refine_iter; itr++) {
                /* r = b - K_unreg * sol */
                P_mul(n, Pp, Pi, Px, sol, r);
                for (int j = 0; j < n; j++) { double s = 0; for (int p = Ap[j]; p < Ap[j + 1]; p++) { int k = rowmap[Ai[p]]; if (k >= 0) s += Ax[p] * sol[n + k]; } r[j] = b[j] - r[j] - s; }
                for (int k = 0; k < nr; k++) r[n + k] = b[n + k];
                for (int j = 0; j < n; j++) for (int p = Ap[j]; p < Ap[j + 1]; p++) { int k = rowmap[Ai[p]]; if (k >= 0) r[n + k] -= Ax[p] * sol[j]; }
                kkt_solve(R, r);
                for (int k = 0; k < NR; k++) sol[k] += r[k];
            }
            double *xpol = sol, *zpol = (double *)malloc(sizeof(double) * (m + 1)), *ypol = (double *)calloc(m + 1, sizeof(double));
            A_mul(n, m, Ap, Ai, Ax, xpol, Axv);
            for (int i = 0; i < m; i++) { double v = Axv[i]; if (v < l[i]) v = l[i]; if (v > u[i]) v = u[i]; zpol[i] = v; }
            for (int k = 0; k < nr; k++) ypol[rows[k]] = sol[n + k];
            for (int i = 0; i < m; i++) tm[i] = Axv[i] - zpol[i];
            double pr = vinf_scaled(Einv, tm, m);
            P_mul(n, Pp, Pi, Px, xpol, Pxv); At_mul(n, Ap, Ai, Ax, ypol, Aty);
            for (int j = 0; j < n; j++) tn[j] = Pxv[j] + q[j] + Aty[j];
            double dr = cinv * vinf_scaled(Dinv, tn, n);
            int ok = (pr < pri_res && dr < dua_res) || (pr < pri_res && dua_res < 1e-10) || (dr < dua_res && pri_res < 1e-10);
            if (ok) { memcpy(x, xpol, sizeof(double) * n); memcpy(z, zpol, sizeof(double) * m); memcpy(y, ypol, sizeof(double) * m); pri_res = pr; dua_res = dr; status_polish = 1; }
            else status_polish = -1;
            free(b); free(sol); free(r); free(rowmap); free(zpol); free(ypol);
        } else status_polish = -1;
        kkt_free(R); free(rows); free(dv);
    }

    /* ---- unscale and report ---- */
    P_mul(n, Pp, Pi, Px, x, Pxv);
    double obj = 0; for (int j = 0; j < n; j++) obj += 0.5 * x[j] * Pxv[j] + q[j] * x[j]; obj *= cinv;
    for (int j = 0; j < n; j++) x_out[j] = D[j] * x[j];
    for (int i = 0; i < m; i++) { z_out[i] = Einv[i] * z[i]; y_out[i] = cinv * E[i] * y[i]; }
    if (info) { info->iter = iter; info->status = status; info->status_polish = status_polish; info->rho_updates = rho_updates;
                info->obj_val = obj; info->pri_res = pri_res; info->dua_res = dua_res; info->rho_estimate = rho; }

    kkt_free(S);
    free(Pp); free(Pi); free(Px); free(Ax); free(q); free(l); free(u); free(D); free(E); free(Dinv); free(Einv); free(dt); free(et);
    free(ctype); free(rho_vec); free(rho_inv); free(kd); free(allrows);
    free(x); free(z); free(y); free(xp); free(zp); free(rhs); free(dx); free(dy); free(Axv); free(Pxv); free(Aty); free(tn); free(tm);
    return status;
}

/* Batched convenience: nb problems sharing the sparsity pattern (values differ). Sequential. */
int oq_solve_batch(int nb, int n, int m, const int *Pp, const int *Pi, const double *Px /*nb x nnzP*/, const double *q /*nb x n*/,
                   const int *Ap, const int *Ai, const double *Ax /*nb x nnzA*/, const double *l, const double *u,
                   const int *perm, const oq_settings *st, double *x, double *y, double *z, oq_info *infos) {
    int nnzP = Pp[n], nnzA = Ap[n], bad = 0;
    for (int b = 0; b < nb; b++) {
        int s = oq_solve(n, m, Pp, Pi, Px + (size_t)b * nnzP, q + (size_t)b * n, Ap, Ai, Ax + (size_t)b * nnzA,
                         l + (size_t)b * m, u + (size_t)b * m, perm, st, x + (size_t)b * n, y + (size_t)b * m, z + (size_t)b * m, infos ? infos + b : NULL);
        if (s != 1) bad++;
    }
    return bad;
}
