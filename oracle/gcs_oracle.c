/*
 * gcs_oracle.c -- CPU restatement (plain C, fp64) of the per-iteration loop of the
 * reference's "full vertex split" ADMM solver.  TEST INFRASTRUCTURE ONLY: it is
 * the checker for the HIP path (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  Nothing shipped may call into this file.
 *
 * What it restates, by reference file:line (/root/reference):
 *   admm_solver_v3.py:352-466  vertex sub-problem (cost :380-413, constraints 1-7 :415-464)
 *   admm_solver_v3.py:469-540  one sub-problem per vertex, solutions scattered into x
 *   admm_solver_v3.py:543-587  edge update = average of the two vertex copies
 *   admm_solver_v3.py:590-614  dual update, primal/dual residual, eps_pri / eps_dual
 *   admm_solver_v3.py:621-733  loop order, rho adaptation (it < 100), stop test
 *   GCS_utils.py:184-211       compute_cost
 *   utils.py:12-28, 85-98      s / t are points (boxes of half-width 1e-6); delta only for 's','t'
 *
 * The reference hands each sub-problem to MOSEK through Drake (neither exists
 * here); this file solves the same convex program with its own primal-dual
 * interior-point method on the reduced "arrow" form:
 * border (x_v, z_v, y_v, t) plus one block (O_e, y_e) of 2n+1 unknowns per
 * incident edge.  Pinned against the reference's committed result records
 * (tests/golden/benchmark{1,2,3,4}.json: stop iteration exact, residual traces,
 * cost) and against tests/ref_dense.py (full, unreduced form) -- see
 * tests/test_oracle_golden.py.
 *
 * Layout shared with the HIP path (include/gcsadmm.h): coupled words per
 * (edge, endpoint) copy are [z_{e,u}[:n], z_{e,w}[:n], y_e] (c = 2n+1 words);
 * copy / mu are [word][incidence] (NI = 2E), zedge is [word][edge].
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 8
#define MAXNW (2 * MAXN + 1)
#define MAXNB (4 * MAXN + 2)

typedef struct {
    int n, V, E;
    const int *edge_tail, *edge_head;
    const int *inc_ptr, *inc_edge, *inc_out;
    const int *edge_inc_tail, *edge_inc_head;
    const int *poly_ptr;
    const double *poly_A;   /* [sum m][n] */
    const double *poly_b;   /* [sum m]   (un-centred) */
    const double *center;   /* [V][n] strictly interior point of each polytope */
    int src, dst;
    /* vertex partitions (all optional): NI = number of copy/mu columns (0 = 2E; > inc_ptr[V] adds ghost
     * columns for the remote endpoint of cut edges), ownership masks for the five norms (NULL = all 1),
     * and the global lengths of the reference's x / mu vectors (0 = from V, E) */
    int NI;
    const unsigned char *inc_counted, *edge_counted;
    double nx_global, nmu_global;
} oracle_graph;

typedef struct {
    double eps_edge;     /* 1e-4, admm_solver_v3.py:388 */
    double ipm_tol;      /* barrier parameter at which the inner solver stops */
    int ipm_max_iter;
    /* warm start of the vertex solves (NULL = every solve starts cold): one record per vertex, warm + warm_ptr[v],
     * oracle_warm_doubles() doubles long, zero-initialised by the caller (valid flag 0) */
    double *warm;
    const long long *warm_ptr;
} oracle_inner_params;

/* ------------------------------------------------------------------ warm start of a vertex solve
 * MOSEK is started cold at admm_solver_v3.py:490; the minimiser it returns does not depend on the start, so neither
 * do the iterates of the ADMM loop beyond the inner tolerance.  The feasible set of a sub-problem does not depend on
 * the targets T or on rho (only the objective does), so an interior iterate of the previous ADMM iteration's solve
 * is a strictly feasible start for the next one.  Rule (the same in the HIP programs, csrc/warm_start.h):
 *   record  = the first iterate, after at least one Newton step, whose barrier parameter is <= WS_SAVE * mu_ref
 *             (primal, multipliers, duals), with rho and the targets of that solve;
 *   restart = from the record when it is valid, rho is unchanged and dT = rho * max|T - T_record| <= theta_v;
 *             mu_ref = max(WS_MU_MIN, WS_KAPPA * dT)  (cold solves: WS_COLD_REF);
 *             the cone pair is re-centred in closed form at mu_ref:  t^2 - mu_ref t - |u|^2 = 0,  lam = (1, -u / t);
 *             the FIRST iteration of a warm solve is a plain Newton step towards s o lam = mu_ref e (no predictor, no
 *             second-order term, no stop test): it re-centres the old iterate on the new problem's central path at a
 *             barrier parameter matched to how far the targets moved; Mehrotra's iterations follow as usual;
 *   a warm solve that fails is repeated cold in the same call;
 *   theta_v = the vertex's own far-warm threshold, kept in its record (starts at WS_COLD_DT, never below WS_NEAR, never above
 *             WS_THETA_MAX) beside n_cold, the iterations of its last cold solve: a solve that started cold because the targets
 *             had moved too far raises theta_v by WS_GROW; a warm solve from dT > WS_NEAR that took more iterations than n_cold,
 *             or failed, lowers it to WS_SHRINK * dT.  Whether a far record beats a cold start depends on the vertex (benchmark4:
 *             warm from dT in [0.3, 1) 12.6 iterations against 9.5 cold; a 40 x 40 lattice: 6.8 against 9.3), and the launch
 *             waits for the slowest solve.
 * Measured on the oracle (tests/warm_rule_sweep.py; sum over the ADMM iterations of the slowest solve): benchmark4 cold 6 370,
 * fixed threshold 0.1 / 0.3 / 1.0: 3 551 / 3 591 / 3 663, per-vertex threshold 3 518; 40 x 40 lattice (200 iterations) 2 727 /
 * 2 151 / 2 074 and 2 071.  Stop iterations 39 / 100 / 508 / 465 unchanged. */
#define WS_NEAR 0.1
#define WS_GROW 1.25
#define WS_SHRINK 0.5
#define WS_THETA_MAX 10.0
static int ws_adapt = 1;                                                  /* (variables only so that tests/warm_rule_sweep.py can sweep the rule) */
static double ws_kappa = 3e-3, ws_mu_min = 1e-7, ws_cold_dt = 1.0;
void oracle_set_warm_adapt(int a) { ws_adapt = a; }
void oracle_set_warm_rule(double kappa, double mu_min, double cold_dt) { ws_kappa = kappa; ws_mu_min = mu_min; ws_cold_dt = cold_dt; }
#define WS_KAPPA ws_kappa
#define WS_MU_MIN ws_mu_min
#define WS_COLD_DT ws_cold_dt
/* how the last vertex solve started (tests/warm_rule_sweep.py): 0 no record / rho changed, 1 cold by dT, 2 warm, 3 warm failed then cold */
static __thread int dbg_kind; static __thread double dbg_dt, dbg_rd;      /* how the last vertex solve started: 0 no record / rho changed, 1 cold by dT, 2 warm, 3 warm failed then cold */
#define WS_SAVE 10.0
#define WS_COLD_REF 1e-4
long long oracle_warm_doubles(int n, int m, int d)
{   /* header (valid, rho, theta, n_cold) | beta | nu | lyv | lsoc | l1 l2 | per block: O, y, l5, l6, l3, l4, targets */
    const int NW = 2 * n + 1, NX = 2 * n, NB = 4 * n + 2, R = 2 * m;
    return 4 + NB + 2 * NW + 2 + (n + 1) + 2 * R + (long long)d * (NX + 3 + 2 * R + NW);
}

/* ------------------------------------------------------------------ small dense helpers */
#define CHOL_SKIP 1e-12
/* Tikhonov term (REG_DELTA/2)|w|^2 on every (centred) unknown: the reference's sub-problem is flat in
 * x_v, z_v, y_v and in the unpenalised halves of the incoming blocks; the term makes the minimiser
 * unique in all components at a bias of ~1e-7, far below the interior-point accuracy. */
#define REG_DELTA 1e-7
static int chol(int n, double *A, int lda)
{   /* lower Cholesky in place.  Late in the interior-point iteration a block Hessian is
     * (huge) x (few active facets) + (tiny) x (rest); a pivot that has cancelled to below CHOL_SKIP
     * of its own diagonal entry is round-off, not curvature: it is clamped to that floor (a
     * continuous rule, so two implementations that differ in the last bits stay close). */
    int frozen = 0;
    double diag[MAXNB];
    for (int j = 0; j < n; ++j) diag[j] = A[j * lda + j];
    for (int j = 0; j < n; ++j) {
        double d = A[j * lda + j];
        for (int k = 0; k < j; ++k) d -= A[j * lda + k] * A[j * lda + k];
        if (!(d > CHOL_SKIP * diag[j])) { d = diag[j] > 0 ? CHOL_SKIP * diag[j] : 1.0; ++frozen; }
        d = sqrt(d);
        A[j * lda + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * lda + j];
            for (int k = 0; k < j; ++k) s -= A[i * lda + k] * A[j * lda + k];
            A[i * lda + j] = s / d;
        }
    }
    return frozen * 0;
}
static void chol_solve(int n, const double *L, int lda, double *b)
{
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * lda + k] * b[k];
        b[i] = s / L[i * lda + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * lda + i] * b[k];
        b[i] = s / L[i * lda + i];
    }
}
/* X (n x n, lda) := inverse of the SPD matrix whose Cholesky factor is L */
static void chol_inverse(int n, const double *L, int lda, double *X)
{
    for (int j = 0; j < n; ++j) {
        double e[MAXNB];
        for (int i = 0; i < n; ++i) e[i] = (i == j);
        chol_solve(n, L, lda, e);
        for (int i = 0; i < n; ++i) X[i * lda + j] = e[i];
    }
}

/* ------------------------------------------------------------------ second-order cone of dimension q */
static double soc_det(int q, const double *s)
{
    double nn = 0;
    for (int k = 1; k < q; ++k) nn += s[k] * s[k];
    nn = sqrt(nn);
    return (s[0] - nn) * (s[0] + nn);
}
/* Nesterov-Todd scaling: W lam = W^{-1} s.  W, Winv are q x q (row-major, ld MAXN+1) */
static int soc_scaling(int q, const double *s, const double *z, double *W, double *Winv, double *wb_out, double *eta_out)
{
    const int ld = MAXN + 1;
    double ss = soc_det(q, s), zz = soc_det(q, z);
    if (!(ss > 0.0) || !(zz > 0.0)) return 1;
    double is = 1.0 / sqrt(ss), iz = 1.0 / sqrt(zz);
    double dot = 0;
    for (int k = 0; k < q; ++k) dot += (s[k] * is) * (z[k] * iz);
    double gam = sqrt(0.5 * (1.0 + dot));
    double wb[MAXN + 1];
    wb[0] = (s[0] * is + z[0] * iz) / (2 * gam);
    for (int k = 1; k < q; ++k) wb[k] = (s[k] * is - z[k] * iz) / (2 * gam);
    double eta = sqrt(sqrt(ss / zz));
    for (int k = 0; k < q; ++k) wb_out[k] = wb[k];
    *eta_out = eta;
    for (int i = 0; i < q; ++i)
        for (int j = 0; j < q; ++j) {
            double w;
            if (i == 0 && j == 0) w = wb[0];
            else if (i == 0) w = wb[j];
            else if (j == 0) w = wb[i];
            else w = (i == j ? 1.0 : 0.0) + wb[i] * wb[j] / (1.0 + wb[0]);
            W[i * ld + j] = eta * w;
            Winv[i * ld + j] = (((i == 0) != (j == 0)) ? -w : w) / eta;
        }
    return 0;
}
static void soc_prod(int q, const double *a, const double *b, double *o)
{
    double d = 0;
    for (int k = 0; k < q; ++k) d += a[k] * b[k];
    o[0] = d;
    for (int k = 1; k < q; ++k) o[k] = a[0] * b[k] + b[0] * a[k];
}
static void soc_div(int q, const double *l, const double *d, double *x)
{   /* l o x = d */
    double det = soc_det(q, l), ld1 = 0;
    for (int k = 1; k < q; ++k) ld1 += l[k] * d[k];
    x[0] = (l[0] * d[0] - ld1) / det;
    for (int k = 1; k < q; ++k) x[k] = (d[k] - x[0] * l[k]) / l[0];
}
static double soc_max_step(int q, const double *s, const double *ds)
{
    double a = ds[0] * ds[0], b = s[0] * ds[0], c = soc_det(q, s);
    for (int k = 1; k < q; ++k) { a -= ds[k] * ds[k]; b -= s[k] * ds[k]; }
    b *= 2;
    double al = 1e300;
    if (ds[0] < 0) al = fmin(al, -s[0] / ds[0]);
    if (fabs(a) < 1e-300) {
        if (b < 0) al = fmin(al, -c / b);
    } else {
        double disc = b * b - 4 * a * c;
        if (disc >= 0) {
            double sq = sqrt(disc);
            double qq = -0.5 * (b + (b >= 0 ? sq : -sq));
            double r1 = qq / a, r2 = (qq != 0.0) ? c / qq : 1e300;
            if (r1 > 0) al = fmin(al, r1);
            if (r2 > 0) al = fmin(al, r2);
        }
    }
    return al;
}
static int soc_interior(int q, const double *s)
{
    double nn = 0;
    for (int k = 1; k < q; ++k) nn += s[k] * s[k];
    return s[0] > sqrt(nn);
}

/* ------------------------------------------------------------------ one incident-edge block */
typedef struct {
    int out;                       /* 1: v is the tail (outgoing edge), both halves penalised */
    double T1[MAXN], T2[MAXN], Ty; /* targets of O[:n], O[n:], y (un-centred coordinates) */
    double O[2 * MAXN], y;         /* centred primal */
    double *l3, *l4;               /* duals of rows 3 / 4, index i*m+j */
    double l5, l6;
    double *s3, *s4, s5, s6;
    double *k3, *k4, k5, k6;       /* per-row right-hand-side multipliers kappa */
    double Kee[MAXNW * MAXNW];     /* factor of the block Hessian */
    double X[MAXNW * 2 * MAXN];    /* K_{omega,x} */
    double B[MAXNW * MAXNW];       /* Kee^{-1} */
    double BX[MAXNW * 2 * MAXN];   /* Kee^{-1} X */
    double g[MAXNW];
    double dw[MAXNW], dwa[MAXNW];
} block_t;

typedef struct {
    int n, m, d, d_in;
    const double *A, *b; /* b centred */
    const double *cen;
    double rho, eps_edge;
    /* border primal: x1,x2,z1,z2,yv,t */
    double beta[MAXNB];
    double nu[2][MAXNW];
    double *l1, *l2, lyv[2], lsoc[MAXN + 1];
    double *s1, *s2, syv[2], ssoc[MAXN + 1];
    double *k1, *k2, kyv[2], ksoc[MAXN + 1];
    block_t *blk;
} vtx_t;

/* gradient of the Lagrangian with the cone multipliers replaced by kappa:
 * gb (border, NB), blk[e].g (block).  With kappa = lambda this is the dual residual. */
static void lagr_grad(vtx_t *P, double *gb)
{
    const int n = P->n, m = P->m, NB = 4 * n + 2, NW = 2 * n + 1;
    const double *A = P->A, *b = P->b;
    for (int k = 0; k < NB; ++k) gb[k] = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < m; ++j) {
            double k1 = P->k1[i * m + j], k2 = P->k2[i * m + j];
            for (int k = 0; k < n; ++k) {
                gb[i * n + k] += A[j * n + k] * k2;
                gb[2 * n + i * n + k] += A[j * n + k] * (k1 - k2);
            }
            gb[4 * n] += b[j] * (k2 - k1);
        }
    gb[4 * n] += -P->kyv[0] + P->kyv[1];
    for (int k = 0; k < n; ++k) {
        gb[2 * n + k] += -P->ksoc[1 + k];
        gb[3 * n + k] += P->ksoc[1 + k];
    }
    gb[4 * n + 1] = 1.0 - P->ksoc[0];
    for (int k = 0; k < NW; ++k) gb[2 * n + k] += P->nu[0][k] + P->nu[1][k];
    for (int k = 0; k < NB - 1; ++k) gb[k] += REG_DELTA * P->beta[k];
    for (int e = 0; e < P->d; ++e) {
        block_t *B = &P->blk[e];
        double *g = B->g;
        const double *nu = P->nu[B->out];
        double gy = 0;
        for (int k = 0; k < n; ++k) {
            double g1 = P->rho * (B->O[k] + B->y * P->cen[k] - B->T1[k]);
            double g2 = B->out ? P->rho * (B->O[n + k] + B->y * P->cen[k] - B->T2[k]) : 0.0;
            g[k] = g1; g[n + k] = g2;
            gy += P->cen[k] * (g1 + g2);
        }
        g[2 * n] = P->rho * (B->y - B->Ty) + P->eps_edge + gy + REG_DELTA * B->y;
        for (int k = 0; k < 2 * n; ++k) g[k] += REG_DELTA * B->O[k];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < m; ++j) {
                double k3 = B->k3[i * m + j], k4 = B->k4[i * m + j];
                for (int k = 0; k < n; ++k) {
                    g[i * n + k] += A[j * n + k] * (k3 - k4);
                    gb[i * n + k] += A[j * n + k] * k4;
                }
                g[2 * n] += b[j] * (k4 - k3);
            }
        g[2 * n] += -B->k5 + B->k6;
        for (int k = 0; k < NW; ++k) g[k] -= nu[k];
    }
}

typedef struct {
    /* factorisation products of one interior-point iteration */
    double Bs[2][MAXNW * MAXNW];   /* sum of B_e per side, then its Cholesky factor */
    double Bsi[2][MAXNW * MAXNW];  /* inverse of that sum */
    double BXs[2][MAXNW * 2 * MAXN];
    double Ys[2][MAXNW * 2 * MAXN]; /* Bs^{-1} BXs */
    double M[MAXNB * MAXNB];       /* reduced border matrix, then its factor */
    double Kb[MAXNB * MAXNB];      /* border Hessian from border rows + blocks' x-part */
    double Wsoc[(MAXN + 1) * (MAXN + 1)], Wsoci[(MAXN + 1) * (MAXN + 1)];
    /* the epigraph variable t is eliminated by hand: W^{-2} = [c0 cv'; cv C11], Su = C11 - cv cv'/c0 */
    double soc_c0, soc_cv[MAXN], soc_Su[MAXN * MAXN];
} fact_t;

/* solve the Newton system for the right-hand side  (-gb, -blk.g, -rp[0], -rp[1]);
 * results: db (border), dnu[2], blk[e].dw */
static void newton_solve(vtx_t *P, fact_t *F, const double *gb, const double rp[2][MAXNW],
                         double *db, double dnu[2][MAXNW])
{
    const int n = P->n, NB = 4 * n + 2, NW = 2 * n + 1, NX = 2 * n;
    double Bg[2][MAXNW], XBg[2 * MAXN];
    memset(Bg, 0, sizeof(Bg)); memset(XBg, 0, sizeof(XBg));
    for (int e = 0; e < P->d; ++e) {
        block_t *B = &P->blk[e];
        double t[MAXNW];
        for (int i = 0; i < NW; ++i) {
            double s = 0;
            for (int k = 0; k < NW; ++k) s += B->B[i * NW + k] * (-B->g[k]);
            t[i] = s; Bg[B->out][i] += s;
        }
        for (int c = 0; c < NX; ++c) {
            double s = 0;
            for (int i = 0; i < NW; ++i) s += B->X[i * NX + c] * t[i];
            XBg[c] += s;
        }
    }
    double rhs[MAXNB];
    for (int k = 0; k < NB; ++k) rhs[k] = -gb[k];
    for (int c = 0; c < NX; ++c) rhs[c] -= XBg[c];
    double u[2][MAXNW];
    for (int s = 0; s < 2; ++s) {
        for (int i = 0; i < NW; ++i) u[s][i] = rp[s][i] - Bg[s][i];
        /* rhs_x -= BXs' Bs^{-1} u ;  rhs_zeta -= Bs^{-1} u */
        double v[MAXNW];
        for (int i = 0; i < NW; ++i) {
            double a = 0;
            for (int k = 0; k < NW; ++k) a += F->Bsi[s][i * NW + k] * u[s][k];
            v[i] = a;
        }
        for (int c = 0; c < NX; ++c) {
            double a = 0;
            for (int i = 0; i < NW; ++i) a += F->BXs[s][i * NX + c] * v[i];
            rhs[c] -= a;
        }
        for (int i = 0; i < NW; ++i) rhs[NX + i] -= v[i];
    }
    {   /* t eliminated: c0 dt + cv'(dz1 - dz2) = -gb[t] */
        const double gt = gb[NB - 1];
        for (int k = 0; k < n; ++k) {
            rhs[2 * n + k] += F->soc_cv[k] * gt / F->soc_c0;
            rhs[3 * n + k] -= F->soc_cv[k] * gt / F->soc_c0;
        }
        rhs[NB - 1] = 0.0;
        /* right-hand side in the (u, z_2) variables: r_u = r_z1, r_z2' = r_z1 + r_z2 */
        for (int k = 0; k < n; ++k) rhs[3 * n + k] += rhs[2 * n + k];
    }
    chol_solve(NB, F->M, NB, rhs);
    for (int k = 0; k < n; ++k) rhs[2 * n + k] += rhs[3 * n + k];   /* dz_1 = du + dz_2 */
    for (int k = 0; k < NB; ++k) db[k] = rhs[k];
    {
        double a = -gb[NB - 1];
        for (int k = 0; k < n; ++k) a -= F->soc_cv[k] * (db[2 * n + k] - db[3 * n + k]);
        db[NB - 1] = a / F->soc_c0;
    }
    for (int s = 0; s < 2; ++s) {
        double w[MAXNW];
        for (int i = 0; i < NW; ++i) {
            double a = db[NX + i] + u[s][i];
            for (int c = 0; c < NX; ++c) a += F->BXs[s][i * NX + c] * db[c];
            w[i] = a;
        }
        for (int i = 0; i < NW; ++i) {
            double a = 0;
            for (int k = 0; k < NW; ++k) a += F->Bsi[s][i * NW + k] * w[k];
            dnu[s][i] = a;
        }
    }
    for (int e = 0; e < P->d; ++e) {
        block_t *B = &P->blk[e];
        double r[MAXNW];
        for (int i = 0; i < NW; ++i) {
            double a = -B->g[i] + dnu[B->out][i];
            for (int c = 0; c < NX; ++c) a -= B->X[i * NX + c] * db[c];
            r[i] = a;
        }
        for (int i = 0; i < NW; ++i) {
            double a = 0;
            for (int k = 0; k < NW; ++k) a += B->B[i * NW + k] * r[k];
            B->dw[i] = a;
        }
    }
}

/* slack increments for a direction (db border, blk.dw); writes into the given arrays */
static void slack_dir(const vtx_t *P, const double *db, double *ds1, double *ds2, double dsyv[2],
                      double *dssoc, int e, const double *dw, double *ds3, double *ds4, double ds56[2])
{
    const int n = P->n, m = P->m;
    const double *A = P->A, *b = P->b;
    if (e < 0) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < m; ++j) {
                double az = 0, ax = 0;
                for (int k = 0; k < n; ++k) { az += A[j * n + k] * db[2 * n + i * n + k]; ax += A[j * n + k] * db[i * n + k]; }
                ds1[i * m + j] = b[j] * db[4 * n] - az;
                ds2[i * m + j] = -b[j] * db[4 * n] - (ax - az);
            }
        dsyv[0] = db[4 * n]; dsyv[1] = -db[4 * n];
        dssoc[0] = db[4 * n + 1];
        for (int k = 0; k < n; ++k) dssoc[1 + k] = db[2 * n + k] - db[3 * n + k];
    } else {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < m; ++j) {
                double ao = 0, ax = 0;
                for (int k = 0; k < n; ++k) { ao += A[j * n + k] * dw[i * n + k]; ax += A[j * n + k] * db[i * n + k]; }
                ds3[i * m + j] = b[j] * dw[2 * n] - ao;
                ds4[i * m + j] = -b[j] * dw[2 * n] - (ax - ao);
            }
        ds56[0] = dw[2 * n]; ds56[1] = -dw[2 * n];
    }
}

static void project_simplex(int d, const double *v, double *out)
{   /* Euclidean projection onto {y >= 0, sum y = 1}: threshold tau by insertion sort */
    double u[256];
    for (int i = 0; i < d; ++i) u[i] = v[i];
    for (int i = 1; i < d; ++i) { double x = u[i]; int j = i - 1; while (j >= 0 && u[j] < x) { u[j + 1] = u[j]; --j; } u[j + 1] = x; }
    double css = 0, tau = 0;
    for (int k = 0; k < d; ++k) {
        css += u[k];
        if (u[k] * (k + 1) > css - 1.0) tau = (css - 1.0) / (k + 1);
    }
    for (int i = 0; i < d; ++i) out[i] = v[i] - tau > 0 ? v[i] - tau : 0.0;
}

/* ------------------------------------------------------------------ one vertex sub-problem
 * Tz: targets [c][d] (word-major, local incidence order): zu[n], zw[n], y.
 * out: copies [c][d] in the same layout, xv[2n], zv[2n], yv.  Returns iterations (<0: not converged). */
static __thread int dbg_vertex = -1;
/* per-thread workspace, grown on demand and reused across solves (no allocation in the hot loop) */
static __thread char *ws_buf = NULL;
static __thread size_t ws_cap = 0;
static void *ws_get(size_t bytes)
{
    if (bytes > ws_cap) { free(ws_buf); ws_cap = bytes + bytes / 2 + 4096; ws_buf = (char *)malloc(ws_cap); }
    return ws_buf;
}
#define TW(k, e) T[(k) * d + (e)]
#define CW(k, e) copy[(k) * d + (e)]
/* one inequality row of a Newton step: dl = kappa - lambda - (lambda / s) ds, the step bound and the sums of the step-length model */
#define ROW(sv, lv, dsv, kv, dlout)                                                          \
    do { double D_ = (lv) / (sv); double dl_ = (kv) - (lv) - D_ * (dsv); (dlout) = dl_;        \
         if ((dsv) < 0) { amax = fmin(amax, -(sv) / (dsv)); }                                \
         if (dl_ < 0) { amax = fmin(amax, -(lv) / dl_); }                                    \
         c1 += (sv) * dl_ + (lv) * (dsv); c2 += (dsv) * dl_; } while (0)
/* ------------------------------------------------------------------ a terminal that is a REGION
 * The reference builds 's' / 't' as points (utils.py:12-28), but its vertex update constrains them like any set
 * (admm_solver_v3.py:415-440 with delta_sv / delta_tv at :450-464).  For v = 's' (v = 't' mirrored, live side = incoming):
 *   (6) with y_v <= 1 forces y_v = 1 and y_e = 0 on the incoming side, so O_e = 0 there (rows 3 of a bounded set),
 *   (7) then reads z_v = x_v = sum_{e out} O_e, (6) sum_{e out} y_e = 1;
 *   rows 1 (A x_i <= b), rows 2 (0 <= 0) and rows 4 (A (x_i - O_{e,i}) <= (1 - y_e) b) are sums of rows 3 of the live blocks and
 *   of the two equalities -- redundant; the bounds on y_e follow from rows 3 and sum y = 1.
 * What is left, over the live blocks (O_e in R^{2n}, y_e) and t:
 *   min  t + sum_e [ eps y_e + rho/2 ( |[O_e]_1 - T1_e|^2 + out_e |[O_e]_2 - T2_e|^2 + (y_e - Ty_e)^2 ) ]
 *   s.t. A [O_e]_i <= y_e b  (i = 1, 2),   sum_e y_e = 1,   | sum_e ([O_e]_1 - [O_e]_2) | <= t.
 * (A point collapses this to the simplex projection above.)  Same primal-dual method as the generic vertex (Mehrotra, one step
 * length, NT scaling of the cone, sigma = (mu_aff / mu)^3, stop on mu), started cold from y_e = 1 / L, O_e = y_e (c, c) or warm from the
 * terminal's record (below).  PARITY UNPINNED by the reference's outputs: no record or fixture of the reference has a terminal with an extent;
 * this restatement is pinned to the reference's formulation as written (tests/test_terminal_region.py: SLSQP on every variable and row of
 * admm_solver_v3.py:352-466 with delta = 1).
 * The Hessian is block diagonal plus the cone and the equality through F = [I, -I, 0; 0, 0, 1]: blocks are eliminated onto
 * (du, dnu), n + 1 unknowns.  The HIP twin is csrc/terminal_region.h. */
static double terminal_extent(int n, int m, const double *A, const double *b_raw, const double *cen)
{   /* the rule of gcsadmm_create: widest distance from the centre to a facet */
    double ext = 0;
    for (int j = 0; j < m; ++j) {
        double nrm = 0, sl = b_raw[j];
        for (int k = 0; k < n; ++k) { nrm += A[j * n + k] * A[j * n + k]; sl -= A[j * n + k] * cen[k]; }
        ext = fmax(ext, fabs(sl) / sqrt(nrm > 0 ? nrm : 1.0));
    }
    return ext;
}
static int gauss_solve(int N, double *M, int ld, double *x)
{   /* in place, partial pivoting; x: right-hand side -> solution */
    for (int c = 0; c < N; ++c) {
        int piv = c;
        for (int r = c + 1; r < N; ++r) if (fabs(M[r * ld + c]) > fabs(M[piv * ld + c])) piv = r;
        if (M[piv * ld + c] == 0.0) return 1;
        if (piv != c) { for (int k = 0; k < N; ++k) { double tmp = M[c * ld + k]; M[c * ld + k] = M[piv * ld + k]; M[piv * ld + k] = tmp; } double tmp = x[c]; x[c] = x[piv]; x[piv] = tmp; }
        for (int r = c + 1; r < N; ++r) {
            const double f = M[r * ld + c] / M[c * ld + c];
            for (int k = c; k < N; ++k) M[r * ld + k] -= f * M[c * ld + k];
            x[r] -= f * x[c];
        }
    }
    for (int r = N - 1; r >= 0; --r) { double a = x[r]; for (int k = r + 1; k < N; ++k) a -= M[r * ld + k] * x[k]; x[r] = a / M[r * ld + r]; }
    return 0;
}
static int solve_terminal_region(int n, int m, const double *A, const double *b, const double *cen, int d, int d_in, int is_src,
                                 const double *T, double rho, const oracle_inner_params *ip, double *copy, double *xv, double *zv, double *yv, double *warm)
{
    const int NW = 2 * n + 1, R = 2 * m, q = n + 1, NF = n + 1, ldq = MAXN + 1;
    const int lo = is_src ? d_in : 0, hi = is_src ? d : d_in, L = hi - lo;
    if (L <= 0) return -2;
    /* per block: p, tg, qd, rhs, hr, dp [NW each] | s, lam, kap, ds, dl [R each] | H [NW*NW] | X [NW*NF] */
    const size_t per = (size_t)6 * NW + 5 * R + (size_t)NW * NW + (size_t)NW * NF;
    double *W = (double *)ws_get(per * L * sizeof(double));
#define TB(e) (W + (size_t)(e) * per)
#define Tp(e) (TB(e))
#define Ttg(e) (TB(e) + NW)
#define Tqd(e) (TB(e) + 2 * NW)
#define Trhs(e) (TB(e) + 3 * NW)
#define Thr(e) (TB(e) + 4 * NW)
#define Tdp(e) (TB(e) + 5 * NW)
#define Ts(e) (TB(e) + 6 * NW)
#define Tlam(e) (Ts(e) + R)
#define Tkap(e) (Ts(e) + 2 * R)
#define Tds(e) (Ts(e) + 3 * R)
#define Tdl(e) (Ts(e) + 4 * R)
#define TH(e) (Ts(e) + 5 * R)
#define TX(e) (TH(e) + NW * NW)
    for (int e = 0; e < L; ++e) {
        const int ge = lo + e;                              /* local incidence */
        double *p = Tp(e), *tg = Ttg(e), *qd = Tqd(e);
        for (int k = 0; k < n; ++k) {
            tg[k] = is_src ? TW(k, ge) : TW(n + k, ge);     /* target of [O]_1 */
            tg[n + k] = is_src ? TW(n + k, ge) : 0.0;       /* target of [O]_2 (outgoing only) */
            qd[k] = rho; qd[n + k] = is_src ? rho : 0.0;
        }
        tg[2 * n] = TW(2 * n, ge); qd[2 * n] = rho;
    }
    /* warm start, the rule of the vertex programs (WS_*, see the head of this file) with a fixed threshold.  Record of a terminal:
     *   [0] valid [1] rho [2] live blocks [3] nu | per live block: p (NW) | targets (NW) | row duals (R)
     * (t and the cone's dual are re-centred at the restart: not kept).  It fits the record every vertex has (oracle_warm_doubles). */
    const int W_HDR = 4, W_PER = 2 * NW + R;
    int use_warm = 0, it_total = 0;
    double mu_ref = WS_COLD_REF;
    if (warm && warm[0] == 1.0 && warm[1] == rho && (int)warm[2] == L) {
        double dT = 0;
        for (int e = 0; e < L; ++e) {
            const double *w = warm + W_HDR + (size_t)e * W_PER + NW, *tg = Ttg(e), *qd = Tqd(e);
            for (int k = 0; k < NW; ++k) if (qd[k] > 0) dT = fmax(dT, fabs(tg[k] - w[k]));
        }
        dT *= rho;
        if (dT <= WS_COLD_DT) { use_warm = 1; mu_ref = fmax(WS_MU_MIN, WS_KAPPA * dT); }
    }
    double t = 1.0, nu = 0.0, ssoc[MAXN + 1], lsoc[MAXN + 1] = {0}, ksoc[MAXN + 1], dssoc[MAXN + 1], dlsoc[MAXN + 1];
    const int deg = L * R + 1;
    int status, it, stalled, saved;
terminal_restart:
    status = -1; stalled = 0; saved = 0;
    for (int k = 0; k < q; ++k) lsoc[k] = 0;
    if (use_warm) {
        double uu = 0, u[MAXN] = {0};
        for (int e = 0; e < L; ++e) {
            const double *w = warm + W_HDR + (size_t)e * W_PER;
            memcpy(Tp(e), w, sizeof(double) * NW); memcpy(Tlam(e), w + 2 * NW, sizeof(double) * R);
            for (int k = 0; k < n; ++k) u[k] += w[k] - w[n + k];
        }
        nu = warm[3];
        for (int k = 0; k < n; ++k) uu += u[k] * u[k];
        t = 0.5 * (mu_ref + sqrt(mu_ref * mu_ref + 4.0 * uu));      /* cone pair re-centred at mu_ref: t^2 - mu_ref t - |u|^2 = 0 */
        lsoc[0] = 1.0;
        for (int k = 0; k < n; ++k) lsoc[1 + k] = -u[k] / t;
    } else {
        for (int e = 0; e < L; ++e) {
            double *p = Tp(e);
            for (int k = 0; k < n; ++k) { p[k] = cen[k] / L; p[n + k] = cen[k] / L; }
            p[2 * n] = 1.0 / L;
        }
        t = 1.0; nu = 0.0; mu_ref = WS_COLD_REF;
    }
    for (it = 0; it <= ip->ipm_max_iter; ++it) {
        const int first_warm = use_warm && it == 0;       /* the re-centring Newton step of a warm solve */
        int interior = 1;
        double gap = 0;
        for (int k = 1; k < q; ++k) ssoc[k] = 0;
        for (int e = 0; e < L; ++e) {
            const double *p = Tp(e); double *s = Ts(e), *lam = Tlam(e);
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < m; ++j) {
                    double a = p[2 * n] * b[j];
                    for (int k = 0; k < n; ++k) a -= A[j * n + k] * p[i * n + k];
                    s[i * m + j] = a;
                    if (!(a > 0)) interior = 0;
                }
            for (int k = 0; k < n; ++k) ssoc[1 + k] += p[k] - p[n + k];
        }
        ssoc[0] = t;
        if (!interior || !soc_interior(q, ssoc)) { status = -3; break; }
        if (it == 0 && !use_warm) {
            for (int e = 0; e < L; ++e) for (int r = 0; r < R; ++r) Tlam(e)[r] = 1.0 / Ts(e)[r];
            lsoc[0] = 1.0 / t;
        }
        for (int e = 0; e < L; ++e) for (int r = 0; r < R; ++r) gap += Ts(e)[r] * Tlam(e)[r];
        for (int k = 0; k < q; ++k) gap += ssoc[k] * lsoc[k];
        const double mu = gap / deg;
        if (warm && !saved && it >= 1 && mu <= WS_SAVE * mu_ref) {      /* the record the next solve of this terminal restarts from */
            saved = 1;
            warm[0] = 1.0; warm[1] = rho; warm[2] = (double)L; warm[3] = nu;
            for (int e = 0; e < L; ++e) {
                double *w = warm + W_HDR + (size_t)e * W_PER;
                memcpy(w, Tp(e), sizeof(double) * NW); memcpy(w + NW, Ttg(e), sizeof(double) * NW); memcpy(w + 2 * NW, Tlam(e), sizeof(double) * R);
            }
        }
        if (!first_warm && (mu <= ip->ipm_tol || (stalled && mu <= 1e3 * ip->ipm_tol))) {
            status = (use_warm && !(mu <= ip->ipm_tol)) ? -7 : 0;       /* (a warm solve does not leave through the precision-exhausted rule) */
            break;
        }
        if (it == ip->ipm_max_iter) break;
        /* scalings */
        double Wsoc[(MAXN + 1) * (MAXN + 1)], Wsoci[(MAXN + 1) * (MAXN + 1)], W2[(MAXN + 1) * (MAXN + 1)], wb[MAXN + 1], eta, lt[MAXN + 1];
        if (soc_scaling(q, ssoc, lsoc, Wsoc, Wsoci, wb, &eta)) { status = (mu <= 1e3 * ip->ipm_tol && !use_warm) ? 0 : -4; break; }
        for (int i = 0; i < q; ++i)
            for (int j = 0; j < q; ++j) { double a = 0; for (int k = 0; k < q; ++k) a += Wsoci[i * ldq + k] * Wsoci[k * ldq + j]; W2[i * ldq + j] = a; }
        for (int i = 0; i < q; ++i) { double a = 0; for (int k = 0; k < q; ++k) a += Wsoc[i * ldq + k] * lsoc[k]; lt[i] = a; }
        /* W^{-2} = [c0 cv'; cv Mu]; t eliminated in closed form (as in the generic vertex): Su = eta^-2 (I - 2 wb1 wb1' / (2 wb0^2 - 1)) */
        const double ie2 = 1.0 / (eta * eta), g2 = 2.0 / (2.0 * wb[0] * wb[0] - 1.0), c0 = ie2 * (2.0 * wb[0] * wb[0] - 1.0);
        double cv[MAXN], Su[MAXN * MAXN];
        for (int k = 0; k < n; ++k) cv[k] = -ie2 * 2.0 * wb[0] * wb[1 + k];
        for (int k = 0; k < n; ++k) for (int l = 0; l < n; ++l) Su[k * n + l] = ie2 * ((k == l ? 1.0 : 0.0) - g2 * wb[1 + k] * wb[1 + l]);
        /* block Hessians, factors, X = H^{-1} F', S = sum F X */
        double S[(MAXN + 1) * (MAXN + 1)] = {0};
        for (int e = 0; e < L; ++e) {
            double *H = TH(e), *X = TX(e); const double *s = Ts(e), *lam = Tlam(e), *qd = Tqd(e);
            memset(H, 0, sizeof(double) * NW * NW);
            for (int k = 0; k < NW; ++k) H[k * NW + k] = qd[k] + REG_DELTA;
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < m; ++j) {
                    const double D = lam[i * m + j] / s[i * m + j];
                    for (int k = 0; k < n; ++k) {
                        for (int l = 0; l < n; ++l) H[(i * n + k) * NW + i * n + l] += D * A[j * n + k] * A[j * n + l];
                        H[(i * n + k) * NW + 2 * n] -= D * A[j * n + k] * b[j];
                        H[2 * n * NW + i * n + k] -= D * A[j * n + k] * b[j];
                    }
                    H[2 * n * NW + 2 * n] += D * b[j] * b[j];
                }
            chol(NW, H, NW);
            for (int c = 0; c < NF; ++c) {
                double col[MAXNW];
                for (int k = 0; k < NW; ++k) col[k] = 0;
                if (c < n) { col[c] = 1.0; col[n + c] = -1.0; } else col[2 * n] = 1.0;
                chol_solve(NW, H, NW, col);
                for (int k = 0; k < NW; ++k) X[k * NF + c] = col[k];
            }
            for (int a = 0; a < NF; ++a)
                for (int c = 0; c < NF; ++c)
                    S[a * NF + c] += a < n ? X[a * NF + c] - X[(n + a) * NF + c] : X[2 * n * NF + c];
        }
        /* one Newton solve for the multipliers kap (rows) / ksoc (cone) in place of the duals; leaves dp, dt, dnu, ds, dssoc */
        double dt = 0, dnu = 0;
#define TERMINAL_NEWTON()                                                                                                        \
        do {                                                                                                                     \
            const double gt = 1.0 - ksoc[0];                                                                                     \
            double z[MAXN + 1] = {0};                                                                                            \
            for (int e = 0; e < L; ++e) {                                                                                        \
                const double *p = Tp(e), *tg = Ttg(e), *qd = Tqd(e), *kap = Tkap(e); double *rhs = Trhs(e), *hr = Thr(e);          \
                double gy = ip->eps_edge + nu;                                                                                   \
                for (int k = 0; k < NW; ++k) rhs[k] = -(qd[k] * (p[k] - tg[k]) + REG_DELTA * p[k]);                                \
                for (int i = 0; i < 2; ++i)                                                                                      \
                    for (int j = 0; j < m; ++j) {                                                                                \
                        for (int k = 0; k < n; ++k) rhs[i * n + k] -= A[j * n + k] * kap[i * m + j];                               \
                        gy -= b[j] * kap[i * m + j];                                                                             \
                    }                                                                                                            \
                rhs[2 * n] -= gy;                                                                                                \
                for (int k = 0; k < n; ++k) { rhs[k] += ksoc[1 + k] + cv[k] * gt / c0; rhs[n + k] -= ksoc[1 + k] + cv[k] * gt / c0; } \
                for (int k = 0; k < NW; ++k) hr[k] = rhs[k];                                                                     \
                chol_solve(NW, TH(e), NW, hr);                                                                                   \
                for (int c = 0; c < NF; ++c) { double a = 0; for (int k = 0; k < NW; ++k) a += TX(e)[k * NF + c] * rhs[k]; z[c] += a; } \
            }                                                                                                                    \
            double Ms[(MAXN + 1) * (MAXN + 1)], sol[MAXN + 1];                                                                   \
            for (int a = 0; a < NF; ++a) {                                                                                       \
                for (int c = 0; c < n; ++c) { double v = 0; for (int k = 0; k < n; ++k) v += S[a * NF + k] * Su[k * n + c]; Ms[a * NF + c] = v + (a == c ? 1.0 : 0.0); } \
                Ms[a * NF + n] = S[a * NF + n];                                                                                  \
                sol[a] = z[a];                                                                                                   \
            }                                                                                                                    \
            if (gauss_solve(NF, Ms, NF, sol)) { status = -5; goto terminal_done; }                                               \
            double su[MAXN + 1];                                                                                                 \
            for (int k = 0; k < n; ++k) { double v = 0; for (int l = 0; l < n; ++l) v += Su[k * n + l] * sol[l]; su[k] = v; }    \
            su[n] = sol[n]; dnu = sol[n];                                                                                        \
            for (int k = 1; k < q; ++k) dssoc[k] = 0;                                                                            \
            for (int e = 0; e < L; ++e) {                                                                                        \
                double *dp = Tdp(e), *ds = Tds(e); const double *hr = Thr(e), *X = TX(e);                                        \
                for (int k = 0; k < NW; ++k) { double a = hr[k]; for (int c = 0; c < NF; ++c) a -= X[k * NF + c] * su[c]; dp[k] = a; } \
                for (int i = 0; i < 2; ++i)                                                                                      \
                    for (int j = 0; j < m; ++j) { double a = dp[2 * n] * b[j]; for (int k = 0; k < n; ++k) a -= A[j * n + k] * dp[i * n + k]; ds[i * m + j] = a; } \
                for (int k = 0; k < n; ++k) dssoc[1 + k] += dp[k] - dp[n + k];                                                   \
            }                                                                                                                    \
            { double a = -gt; for (int k = 0; k < n; ++k) a -= cv[k] * dssoc[1 + k]; dt = a / c0; }                              \
            dssoc[0] = dt;                                                                                                       \
        } while (0)
        /* predictor (the first iteration of a warm solve has none: kappa = mu_ref / s, no second-order term) */
        double amax = 1e300, c1 = 0, c2 = 0, sm;
        for (int e = 0; e < L; ++e) memset(Tkap(e), 0, sizeof(double) * R);
        for (int k = 0; k < q; ++k) { ksoc[k] = 0; dssoc[k] = 0; dlsoc[k] = 0; }
        if (first_warm) sm = mu_ref;
        else {
        TERMINAL_NEWTON();
        for (int e = 0; e < L; ++e) {
            double *s = Ts(e), *lam = Tlam(e), *ds = Tds(e), *dl = Tdl(e), *kap = Tkap(e);
            for (int r = 0; r < R; ++r) { ROW(s[r], lam[r], ds[r], 0.0, dl[r]); kap[r] = ds[r] * dl[r]; }
        }
        for (int i = 0; i < q; ++i) { double a = -lsoc[i]; for (int k = 0; k < q; ++k) a -= W2[i * ldq + k] * dssoc[k]; dlsoc[i] = a; }
        amax = fmin(amax, fmin(soc_max_step(q, ssoc, dssoc), soc_max_step(q, lsoc, dlsoc)));
        for (int k = 0; k < q; ++k) { c1 += ssoc[k] * dlsoc[k] + lsoc[k] * dssoc[k]; c2 += dssoc[k] * dlsoc[k]; }
        const double al_aff = fmin(1.0, amax);
        double sig = (gap + al_aff * c1 + al_aff * al_aff * c2) / deg / mu;
        sig = sig < 0 ? 0 : (sig > 1 ? 1 : sig); sig = sig * sig * sig;
        sm = sig * mu;
        }
        /* corrector multipliers */
        for (int e = 0; e < L; ++e) { double *kap = Tkap(e); const double *s = Ts(e); for (int r = 0; r < R; ++r) kap[r] = (sm - kap[r]) / s[r]; }
        {
            double a1[MAXN + 1], a2[MAXN + 1], pr[MAXN + 1], qv[MAXN + 1];
            for (int i = 0; i < q; ++i) {
                double u1 = 0, u2 = 0;
                for (int k = 0; k < q && !first_warm; ++k) { u1 += Wsoci[i * ldq + k] * dssoc[k]; u2 += Wsoc[i * ldq + k] * dlsoc[k]; }
                a1[i] = u1; a2[i] = u2;
            }
            soc_prod(q, a1, a2, pr);
            soc_div(q, lt, pr, qv);
            const double dets = soc_det(q, ssoc);
            for (int i = 0; i < q; ++i) {
                double a = 0;
                for (int k = 0; k < q; ++k) a += Wsoci[i * ldq + k] * qv[k];
                ksoc[i] = sm * (i == 0 ? ssoc[0] : -ssoc[i]) / dets - a;
            }
        }
        TERMINAL_NEWTON();
        amax = 1e300; c1 = c2 = 0;
        for (int e = 0; e < L; ++e) {
            double *s = Ts(e), *lam = Tlam(e), *ds = Tds(e), *dl = Tdl(e), *kap = Tkap(e);
            for (int r = 0; r < R; ++r) ROW(s[r], lam[r], ds[r], kap[r], dl[r]);
        }
        for (int i = 0; i < q; ++i) { double a = ksoc[i] - lsoc[i]; for (int k = 0; k < q; ++k) a -= W2[i * ldq + k] * dssoc[k]; dlsoc[i] = a; }
        amax = fmin(amax, fmin(soc_max_step(q, ssoc, dssoc), soc_max_step(q, lsoc, dlsoc)));
        double al = fmin(1.0, 0.99 * amax);
        for (int tries = 0; tries < 40; ++tries) {
            double s2[MAXN + 1] = {0}, l2[MAXN + 1] = {0};
            for (int k = 0; k < q; ++k) { s2[k] = ssoc[k] + al * dssoc[k]; l2[k] = lsoc[k] + al * dlsoc[k]; }
            if (soc_interior(q, s2) && soc_interior(q, l2)) break;
            al *= 0.7;
        }
        stalled = al < 1e-3;
        for (int e = 0; e < L; ++e) {
            double *p = Tp(e), *lam = Tlam(e); const double *dp = Tdp(e), *dl = Tdl(e);
            for (int k = 0; k < NW; ++k) p[k] += al * dp[k];
            for (int r = 0; r < R; ++r) lam[r] += al * dl[r];
        }
        t += al * dt; nu += al * dnu;
        for (int k = 0; k < q; ++k) lsoc[k] += al * dlsoc[k];
    }
terminal_done:
    it_total += it;
    if (status != 0 && warm) warm[0] = 0.0;                                     /* no restart from a solve that failed */
    if (status != 0 && use_warm) { use_warm = 0; goto terminal_restart; }        /* a failed warm solve is repeated cold */
    if (status != 0) return status < -1 ? status : -1;
    for (int k = 0; k < 2 * n; ++k) xv[k] = 0;
    for (int e = 0; e < d; ++e) {
        const int live = e >= lo && e < hi, outgoing = e >= d_in;
        const double *p = live ? Tp(e - lo) : NULL;
        for (int k = 0; k < n; ++k) {
            CW(k, e) = outgoing ? (live ? p[k] : 0.0) : TW(k, e);
            CW(n + k, e) = live ? (outgoing ? p[n + k] : p[k]) : 0.0;
        }
        CW(2 * n, e) = live ? p[2 * n] : 0.0;
        if (live) for (int k = 0; k < 2 * n; ++k) xv[k] += p[k];
    }
    for (int k = 0; k < 2 * n; ++k) zv[k] = xv[k];
    *yv = 1.0;
    return it_total;
}

int oracle_solve_vertex(int n, int m, const double *A, const double *b_raw, const double *cen,
                        int d, int d_in, int is_src, int is_dst, const double *T, double rho,
                        const oracle_inner_params *ip, double *copy, double *xv, double *zv, double *yv, double *warm)
{
    const int NW = 2 * n + 1, NX = 2 * n, NB = 4 * n + 2, q = n + 1, c = 2 * n + 1;
    const int d_out = d - d_in;
    if ((is_src || is_dst) && terminal_extent(n, m, A, b_raw, cen) > 1e-5)
        return solve_terminal_region(n, m, A, b_raw, cen, d, d_in, is_src, T, rho, ip, copy, xv, zv, yv, warm);
    if (is_src || is_dst) {
        /* point vertex (box of half-width 1e-6 around cen): O_{e,i} = y_e * pt on the live side,
         * sum y_e = 1 -> separable quadratic over the simplex; the other side is dead (y = 0). */
        const int lo = is_src ? d_in : 0, hi = is_src ? d : d_in, na = hi - lo;
        double pp = 0;
        for (int k = 0; k < n; ++k) pp += cen[k] * cen[k];
        double a = is_src ? 2 * pp + 1 : pp + 1;
        double v[256], y[256];
        if (na > 256) return -2;
        for (int e = lo; e < hi; ++e) {
            double cc = TW(2 * n, e);
            for (int k = 0; k < n; ++k) cc += cen[k] * (is_src ? TW(k, e) + TW(n + k, e) : TW(n + k, e));
            v[e - lo] = (cc - ip->eps_edge / rho) / a;
        }
        if (na > 0) project_simplex(na, v, y);
        for (int e = 0; e < d; ++e) {
            int live = (e >= lo && e < hi);
            double ye = live ? y[e - lo] : 0.0;
            int outgoing = e >= d_in;
            for (int k = 0; k < n; ++k) {
                if (outgoing) { CW(k, e) = ye * cen[k]; CW(n + k, e) = ye * cen[k]; }
                else { CW(k, e) = TW(k, e); CW(n + k, e) = ye * cen[k]; }
            }
            CW(2 * n, e) = ye;
        }
        for (int k = 0; k < n; ++k) { xv[k] = xv[n + k] = cen[k]; zv[k] = zv[n + k] = cen[k]; }
        *yv = 1.0;
        return 0;
    }
    if (d_in == 0 || d_out == 0) {
        /* no flow can pass: y = 0, z = 0; the only free coupled word (z_{e,a}[:n] of an incoming
         * edge) is penalised only and sits at its target */
        for (int e = 0; e < d; ++e) {
            for (int k = 0; k < n; ++k) { CW(k, e) = e < d_in ? TW(k, e) : 0.0; CW(n + k, e) = 0.0; }
            CW(2 * n, e) = 0.0;
        }
        for (int k = 0; k < n; ++k) { xv[k] = xv[n + k] = cen[k]; zv[k] = zv[n + k] = 0.0; }
        *yv = 0.0;
        return 0;
    }

    /* ---- generic vertex: primal-dual interior point on the centred arrow form ---- */
    vtx_t P; fact_t F;
    P.n = n; P.m = m; P.d = d; P.d_in = d_in; P.A = A; P.cen = cen; P.rho = rho; P.eps_edge = ip->eps_edge;
    const int R = 2 * m;
    const size_t n_pool = (size_t)(6 * R) + (size_t)d * 6 * R, n_ds = (size_t)4 * R + (size_t)d * 4 * R;
    const size_t off_pool = ((size_t)m * sizeof(double) + 63) & ~(size_t)63;
    const size_t off_ds = (off_pool + n_pool * sizeof(double) + 63) & ~(size_t)63;
    const size_t off_blk = (off_ds + n_ds * sizeof(double) + 63) & ~(size_t)63;
    char *ws = (char *)ws_get(off_blk + (size_t)d * sizeof(block_t));
    double *bc = (double *)ws;
    for (int j = 0; j < m; ++j) { double s = b_raw[j]; for (int k = 0; k < n; ++k) s -= A[j * n + k] * cen[k]; bc[j] = s; }
    P.b = bc;
    double *pool = (double *)(ws + off_pool);
    memset(pool, 0, n_pool * sizeof(double));
    P.l1 = pool; P.l2 = pool + R; P.s1 = pool + 2 * R; P.s2 = pool + 3 * R; P.k1 = pool + 4 * R; P.k2 = pool + 5 * R;
    P.blk = (block_t *)(ws + off_blk);
    double *ds1 = (double *)(ws + off_ds);
    double *ds2 = ds1 + R, *dl1 = ds1 + 2 * R, *dl2 = ds1 + 3 * R;
    for (int e = 0; e < d; ++e) {
        block_t *B = &P.blk[e];
        double *p = pool + 6 * R + (size_t)e * 6 * R;
        B->l3 = p; B->l4 = p + R; B->s3 = p + 2 * R; B->s4 = p + 3 * R; B->k3 = p + 4 * R; B->k4 = p + 5 * R;
        B->out = e >= d_in;
        for (int k = 0; k < n; ++k) {
            B->T1[k] = B->out ? TW(k, e) : TW(n + k, e);
            B->T2[k] = B->out ? TW(n + k, e) : 0.0;
        }
        B->Ty = TW(2 * n, e);
    }
    /* warm-start record of this vertex (layout: oracle_warm_doubles) */
    const int W_BETA = 4, W_NU = W_BETA + NB, W_LYV = W_NU + 2 * NW, W_LSOC = W_LYV + 2, W_L1 = W_LSOC + q, W_L2 = W_L1 + R,
              W_BLK = W_L2 + R, W_BS = NX + 3 + 2 * R + NW, W_BT = NX + 3 + 2 * R;
    int use_warm = 0, it_total = 0;
    double mu_ref = WS_COLD_REF;
    double dT = -1.0;        /* < 0: no comparable record */
    if (warm && warm[0] == 1.0 && warm[1] == rho) {
        dT = 0;
        for (int e = 0; e < d; ++e) {
            const block_t *B = &P.blk[e]; const double *w = warm + W_BLK + (size_t)e * W_BS + W_BT;
            for (int k = 0; k < n; ++k) { dT = fmax(dT, fabs(B->T1[k] - w[k])); if (B->out) dT = fmax(dT, fabs(B->T2[k] - w[n + k])); }
            dT = fmax(dT, fabs(B->Ty - w[2 * n]));
        }
        dT *= rho;
        const double theta = ws_adapt ? fmax(WS_NEAR, warm[2] > 0 ? warm[2] : WS_COLD_DT) : WS_COLD_DT;
        if (dT <= theta) { use_warm = 1; mu_ref = fmax(WS_MU_MIN, WS_KAPPA * dT); }
        dbg_kind = use_warm ? 2 : 1; dbg_dt = dT;
    } else dbg_kind = 0;
    const int was_warm = use_warm;
    int status, it, stalled, saved;
    const double mu0 = 1.0;
    double scale = 1.0;
    const int deg = 4 * m + 2 + 1 + d * (4 * m + 2);
restart:
    status = -1; stalled = 0; saved = 0;
    if (use_warm) {
        for (int k = 0; k < NB; ++k) P.beta[k] = warm[W_BETA + k];
        for (int s = 0; s < 2; ++s) for (int k = 0; k < NW; ++k) P.nu[s][k] = warm[W_NU + s * NW + k];
        P.lyv[0] = warm[W_LYV]; P.lyv[1] = warm[W_LYV + 1];
        for (int r = 0; r < R; ++r) { P.l1[r] = warm[W_L1 + r]; P.l2[r] = warm[W_L2 + r]; }
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e]; const double *w = warm + W_BLK + (size_t)e * W_BS;
            for (int k = 0; k < NX; ++k) B->O[k] = w[k];
            B->y = w[NX]; B->l5 = w[NX + 1]; B->l6 = w[NX + 2];
            for (int r = 0; r < R; ++r) { B->l3[r] = w[NX + 3 + r]; B->l4[r] = w[NX + 3 + R + r]; }
        }
        /* cone pair re-centred at mu_ref */
        double uu = 0;
        for (int k = 0; k < n; ++k) { const double u = P.beta[2 * n + k] - P.beta[3 * n + k]; uu += u * u; }
        const double tn = 0.5 * (mu_ref + sqrt(mu_ref * mu_ref + 4.0 * uu));
        P.beta[4 * n + 1] = tn;
        P.lsoc[0] = 1.0;
        for (int k = 0; k < n; ++k) P.lsoc[1 + k] = -(P.beta[2 * n + k] - P.beta[3 * n + k]) / tn;
    } else {
        mu_ref = WS_COLD_REF;
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            for (int k = 0; k < NX; ++k) B->O[k] = 0.0;
            B->y = 0.5 / (B->out ? d_out : d_in);
        }
        for (int k = 0; k < NB; ++k) P.beta[k] = 0.0;
        P.beta[4 * n] = 0.5; P.beta[4 * n + 1] = 1.0;
        memset(P.nu, 0, sizeof(P.nu));
    }
    for (it = 0; it <= ip->ipm_max_iter; ++it) {
        const int first_warm = use_warm && it == 0;   /* the re-centring Newton step of a warm solve */
        /* slacks */
        const double *x = P.beta, *z = P.beta + 2 * n; const double yvv = P.beta[4 * n], t = P.beta[4 * n + 1];
        int interior = 1;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < m; ++j) {
                double az = 0, ax = 0;
                for (int k = 0; k < n; ++k) { az += A[j * n + k] * z[i * n + k]; ax += A[j * n + k] * x[i * n + k]; }
                P.s1[i * m + j] = bc[j] * yvv - az;
                P.s2[i * m + j] = bc[j] * (1 - yvv) - (ax - az);
                interior &= P.s1[i * m + j] > 0 && P.s2[i * m + j] > 0;
            }
        P.syv[0] = yvv; P.syv[1] = 1 - yvv;
        P.ssoc[0] = t;
        for (int k = 0; k < n; ++k) P.ssoc[1 + k] = z[k] - z[n + k];
        interior &= soc_interior(q, P.ssoc);
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < m; ++j) {
                    double ao = 0, ax = 0;
                    for (int k = 0; k < n; ++k) { ao += A[j * n + k] * B->O[i * n + k]; ax += A[j * n + k] * x[i * n + k]; }
                    B->s3[i * m + j] = bc[j] * B->y - ao;
                    B->s4[i * m + j] = bc[j] * (1 - B->y) - (ax - ao);
                    interior &= B->s3[i * m + j] > 0 && B->s4[i * m + j] > 0;
                }
            B->s5 = B->y; B->s6 = 1 - B->y;
        }
        if (!interior) { status = -3; break; }
        if (it == 0 && !use_warm) {
            for (int r = 0; r < R; ++r) { P.l1[r] = mu0 / P.s1[r]; P.l2[r] = mu0 / P.s2[r]; }
            P.lyv[0] = mu0 / P.syv[0]; P.lyv[1] = mu0 / P.syv[1];
            P.lsoc[0] = mu0 / t; for (int k = 1; k < q; ++k) P.lsoc[k] = 0.0;
            for (int e = 0; e < d; ++e) {
                block_t *B = &P.blk[e];
                for (int r = 0; r < R; ++r) { B->l3[r] = mu0 / B->s3[r]; B->l4[r] = mu0 / B->s4[r]; }
                B->l5 = mu0 / B->s5; B->l6 = mu0 / B->s6;
            }
        }
        /* complementarity, residuals */
        double gap = P.syv[0] * P.lyv[0] + P.syv[1] * P.lyv[1];
        for (int k = 0; k < q; ++k) gap += P.ssoc[k] * P.lsoc[k];
        for (int r = 0; r < R; ++r) gap += P.s1[r] * P.l1[r] + P.s2[r] * P.l2[r];
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            for (int r = 0; r < R; ++r) gap += B->s3[r] * B->l3[r] + B->s4[r] * B->l4[r];
            gap += B->s5 * B->l5 + B->s6 * B->l6;
        }
        const double mu = gap / deg;
        if (getenv("GCS_EMU_TRACE")) fprintf(stderr, "v %d it %d mu %.17g\n", dbg_vertex, it, mu);
        /* dual residual: kappa := lambda */
        memcpy(P.k1, P.l1, sizeof(double) * R); memcpy(P.k2, P.l2, sizeof(double) * R);
        P.kyv[0] = P.lyv[0]; P.kyv[1] = P.lyv[1];
        for (int k = 0; k < q; ++k) P.ksoc[k] = P.lsoc[k];
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            memcpy(B->k3, B->l3, sizeof(double) * R); memcpy(B->k4, B->l4, sizeof(double) * R);
            B->k5 = B->l5; B->k6 = B->l6;
        }
        double gb[MAXNB];
        lagr_grad(&P, gb);
        double rdmax = 0;
        for (int k = 0; k < NB; ++k) rdmax = fmax(rdmax, fabs(gb[k]));
        for (int e = 0; e < d; ++e) for (int k = 0; k < NW; ++k) rdmax = fmax(rdmax, fabs(P.blk[e].g[k]));
        double rp[2][MAXNW], rpmax = 0;
        for (int s = 0; s < 2; ++s) for (int k = 0; k < NW; ++k) rp[s][k] = P.beta[2 * n + k];
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            for (int k = 0; k < NX; ++k) rp[B->out][k] -= B->O[k];
            rp[B->out][2 * n] -= B->y;
        }
        for (int s = 0; s < 2; ++s) for (int k = 0; k < NW; ++k) rpmax = fmax(rpmax, fabs(rp[s][k]));
        if (it == 0) {
            /* scale of the objective gradient at the start (the lambda part excluded) */
            for (int e = 0; e < d; ++e) {
                block_t *B = &P.blk[e];
                for (int k = 0; k < n; ++k) {
                    scale = fmax(scale, 1 + fabs(rho * (B->y * cen[k] - B->T1[k])));
                    if (B->out) scale = fmax(scale, 1 + fabs(rho * (B->y * cen[k] - B->T2[k])));
                }
                scale = fmax(scale, 1 + fabs(rho * (B->y - B->Ty)));
            }
        }
        /* stop on the barrier parameter alone: it is the one convergence measure that is insensitive to the
         * round-off of the (ill-conditioned) Newton solves; the residuals shrink at least as fast from the
         * strictly feasible start (rdmax / rpmax are kept for diagnostics) */
        (void)rdmax; (void)rpmax; (void)scale;
        /* a vanishing step means the linear algebra has run out of precision: further iterations cannot
         * improve the point; accept it if the barrier parameter is within 1e3 of the target */
        if (warm && !saved && it >= 1 && mu <= WS_SAVE * mu_ref) {   /* the record the next solve of this vertex restarts from */
            saved = 1;
            warm[0] = 1.0; warm[1] = rho;
            for (int k = 0; k < NB; ++k) warm[W_BETA + k] = P.beta[k];
            for (int s = 0; s < 2; ++s) for (int k = 0; k < NW; ++k) warm[W_NU + s * NW + k] = P.nu[s][k];
            warm[W_LYV] = P.lyv[0]; warm[W_LYV + 1] = P.lyv[1];
            for (int k = 0; k < q; ++k) warm[W_LSOC + k] = P.lsoc[k];
            for (int r = 0; r < R; ++r) { warm[W_L1 + r] = P.l1[r]; warm[W_L2 + r] = P.l2[r]; }
            for (int e = 0; e < d; ++e) {
                const block_t *B = &P.blk[e]; double *w = warm + W_BLK + (size_t)e * W_BS;
                for (int k = 0; k < NX; ++k) w[k] = B->O[k];
                w[NX] = B->y; w[NX + 1] = B->l5; w[NX + 2] = B->l6;
                for (int r = 0; r < R; ++r) { w[NX + 3 + r] = B->l3[r]; w[NX + 3 + R + r] = B->l4[r]; }
                for (int k = 0; k < n; ++k) { w[W_BT + k] = B->T1[k]; w[W_BT + n + k] = B->T2[k]; }
                w[W_BT + 2 * n] = B->Ty;
            }
        }
        if (!first_warm && (mu <= ip->ipm_tol || (stalled && mu <= 1e3 * ip->ipm_tol))) {
            /* a WARM solve does not leave through the precision-exhausted rule: its start was off-centre (an iterate jammed against a
             * bound makes steps of 1e-6 and is several 1e-3 off in that vertex's words); it counts as a failed warm attempt and is
             * repeated cold below.  Cold solves keep the rule (on the fixtures it fires at mu <= 5e-9 only). */
            status = (use_warm && !(mu <= ip->ipm_tol)) ? -7 : 0;
            dbg_rd = rdmax;
            break;
        }
        if (it == ip->ipm_max_iter) break;

        /* ---- scalings, block Hessians, border Hessian ---- */
        double wb[MAXN + 1], eta;
        if (soc_scaling(q, P.ssoc, P.lsoc, F.Wsoc, F.Wsoci, wb, &eta)) { status = (mu <= 1e3 * ip->ipm_tol && !use_warm) ? 0 : -4; break; }
        const int ldq = MAXN + 1;
        double W2[(MAXN + 1) * (MAXN + 1)]; /* W^{-2} */
        for (int i = 0; i < q; ++i)
            for (int j = 0; j < q; ++j) {
                double a = 0;
                for (int k = 0; k < q; ++k) a += F.Wsoci[i * ldq + k] * F.Wsoci[k * ldq + j];
                W2[i * ldq + j] = a;
            }
        double lt[MAXN + 1]; /* scaled variable W lam */
        for (int i = 0; i < q; ++i) { double a = 0; for (int k = 0; k < q; ++k) a += F.Wsoc[i * ldq + k] * P.lsoc[k]; lt[i] = a; }
        memset(F.Kb, 0, sizeof(double) * NB * NB);
#define KB(i, j) F.Kb[(i) * NB + (j)]
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < m; ++j) {
                double D1 = P.l1[i * m + j] / P.s1[i * m + j], D2 = P.l2[i * m + j] / P.s2[i * m + j];
                for (int k = 0; k < n; ++k) {
                    for (int l = 0; l < n; ++l) {
                        double aa = A[j * n + k] * A[j * n + l];
                        KB(2 * n + i * n + k, 2 * n + i * n + l) += (D1 + D2) * aa;
                        KB(i * n + k, i * n + l) += D2 * aa;
                        KB(i * n + k, 2 * n + i * n + l) -= D2 * aa;
                        KB(2 * n + i * n + l, i * n + k) -= D2 * aa;
                    }
                    KB(2 * n + i * n + k, 4 * n) -= (D1 + D2) * bc[j] * A[j * n + k];
                    KB(4 * n, 2 * n + i * n + k) -= (D1 + D2) * bc[j] * A[j * n + k];
                    KB(i * n + k, 4 * n) += D2 * bc[j] * A[j * n + k];
                    KB(4 * n, i * n + k) += D2 * bc[j] * A[j * n + k];
                }
                KB(4 * n, 4 * n) += (D1 + D2) * bc[j] * bc[j];
            }
        KB(4 * n, 4 * n) += P.lyv[0] / P.syv[0] + P.lyv[1] / P.syv[1];
        for (int k = 0; k < NB - 1; ++k) KB(k, k) += REG_DELTA;
        /* Cone block.  With W^{-2} = eta^{-2}(2 v v' - J), v = (wb0, -wb1), eliminating t first leaves on
         * u = z_1 - z_2 the Schur complement  Su = eta^{-2} (I - 2 wb1 wb1' / (2 wb0^2 - 1)),  formed from
         * this closed form: pivoting on t numerically (or last) cancels catastrophically once the cone is
         * active (W^{-2} is then ~1/mu times a rank-one matrix). */
        {
            const double ie2 = 1.0 / (eta * eta), g2 = 2.0 / (2.0 * wb[0] * wb[0] - 1.0);
            F.soc_c0 = ie2 * (2.0 * wb[0] * wb[0] - 1.0);
            for (int k = 0; k < n; ++k) F.soc_cv[k] = -ie2 * 2.0 * wb[0] * wb[1 + k];
            for (int k = 0; k < n; ++k)
                for (int l = 0; l < n; ++l) {
                    const double w = ie2 * ((k == l ? 1.0 : 0.0) - g2 * wb[1 + k] * wb[1 + l]);
                    F.soc_Su[k * n + l] = w;   /* added to M after the change of variables below */
                }
            KB(4 * n + 1, 4 * n + 1) = 1.0;   /* decoupled placeholder: t is recovered after the solve */
        }
        memset(F.Bs, 0, sizeof(F.Bs)); memset(F.BXs, 0, sizeof(F.BXs));
        double XBX[2 * MAXN * 2 * MAXN];
        memset(XBX, 0, sizeof(XBX));
        int bad = 0;
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            double *K = B->Kee;
            memset(K, 0, sizeof(double) * NW * NW); memset(B->X, 0, sizeof(double) * NW * NX);
            for (int k = 0; k < n; ++k) {
                K[k * NW + k] += rho; K[k * NW + 2 * n] += rho * cen[k]; K[2 * n * NW + k] += rho * cen[k];
                K[2 * n * NW + 2 * n] += rho * cen[k] * cen[k];
                if (B->out) {
                    K[(n + k) * NW + n + k] += rho; K[(n + k) * NW + 2 * n] += rho * cen[k]; K[2 * n * NW + n + k] += rho * cen[k];
                    K[2 * n * NW + 2 * n] += rho * cen[k] * cen[k];
                }
            }
            K[2 * n * NW + 2 * n] += rho + B->l5 / B->s5 + B->l6 / B->s6;
            for (int k = 0; k < NW; ++k) K[k * NW + k] += REG_DELTA;
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < m; ++j) {
                    double D3 = B->l3[i * m + j] / B->s3[i * m + j], D4 = B->l4[i * m + j] / B->s4[i * m + j];
                    for (int k = 0; k < n; ++k) {
                        for (int l = 0; l < n; ++l) {
                            double aa = A[j * n + k] * A[j * n + l];
                            K[(i * n + k) * NW + i * n + l] += (D3 + D4) * aa;
                            B->X[(i * n + k) * NX + i * n + l] -= D4 * aa;
                            KB(i * n + k, i * n + l) += D4 * aa;
                        }
                        K[(i * n + k) * NW + 2 * n] -= (D3 + D4) * bc[j] * A[j * n + k];
                        K[2 * n * NW + i * n + k] -= (D3 + D4) * bc[j] * A[j * n + k];
                        B->X[2 * n * NX + i * n + k] += D4 * bc[j] * A[j * n + k];
                    }
                    K[2 * n * NW + 2 * n] += (D3 + D4) * bc[j] * bc[j];
                }
            double Ksave[MAXNW * MAXNW]; memcpy(Ksave, K, sizeof(double) * NW * NW);
            if (chol(NW, K, NW)) {
                bad = 1;
                if (getenv("GCS_ORACLE_DEBUG2")) {
                    fprintf(stderr, "block %d out=%d y=%.3e mu=%.3e l5=%.3e l6=%.3e\n", e, B->out, B->y, mu, B->l5, B->l6);
                    for (int i = 0; i < NW; ++i) { for (int k = 0; k < NW; ++k) fprintf(stderr, " %.6e", Ksave[i * NW + k]); fprintf(stderr, "\n"); }
                    for (int r = 0; r < R; ++r) fprintf(stderr, "  row %d s3=%.3e l3=%.3e s4=%.3e l4=%.3e\n", r, B->s3[r], B->l3[r], B->s4[r], B->l4[r]);
                }
                break;
            }
            chol_inverse(NW, K, NW, B->B);
            for (int i = 0; i < NW; ++i)
                for (int cc = 0; cc < NX; ++cc) {
                    double a = 0;
                    for (int k = 0; k < NW; ++k) a += B->B[i * NW + k] * B->X[k * NX + cc];
                    B->BX[i * NX + cc] = a;
                }
            for (int i = 0; i < NW * NW; ++i) F.Bs[B->out][i] += B->B[i];
            for (int i = 0; i < NW * NX; ++i) F.BXs[B->out][i] += B->BX[i];
            for (int a_ = 0; a_ < NX; ++a_)
                for (int cc = 0; cc < NX; ++cc) {
                    double a = 0;
                    for (int k = 0; k < NW; ++k) a += B->X[k * NX + a_] * B->BX[k * NX + cc];
                    XBX[a_ * NX + cc] += a;
                }
        }
        if (bad) { status = -5; break; }
        memcpy(F.M, F.Kb, sizeof(double) * NB * NB);
        for (int a_ = 0; a_ < NX; ++a_) for (int cc = 0; cc < NX; ++cc) F.M[a_ * NB + cc] -= XBX[a_ * NX + cc];
        for (int s = 0; s < 2; ++s) {
            if (chol(NW, F.Bs[s], NW)) { bad = 1; break; }
            chol_inverse(NW, F.Bs[s], NW, F.Bsi[s]);
            for (int i = 0; i < NW; ++i)
                for (int cc = 0; cc < NX; ++cc) {
                    double a = 0;
                    for (int k = 0; k < NW; ++k) a += F.Bsi[s][i * NW + k] * F.BXs[s][k * NX + cc];
                    F.Ys[s][i * NX + cc] = a;
                }
            for (int a_ = 0; a_ < NX; ++a_)
                for (int cc = 0; cc < NX; ++cc) {
                    double a = 0;
                    for (int k = 0; k < NW; ++k) a += F.BXs[s][k * NX + a_] * F.Ys[s][k * NX + cc];
                    F.M[a_ * NB + cc] += a;
                }
            for (int i = 0; i < NW; ++i)
                for (int cc = 0; cc < NX; ++cc) {
                    F.M[(NX + i) * NB + cc] += F.Ys[s][i * NX + cc];
                    F.M[cc * NB + NX + i] += F.Ys[s][i * NX + cc];
                }
            for (int i = 0; i < NW; ++i)
                for (int k = 0; k < NW; ++k) F.M[(NX + i) * NB + NX + k] += F.Bsi[s][i * NW + k];
        }
        /* Change of variables (u, z_2) = (z_1 - z_2, z_2) in the border system: the cone term then sits on
         * u alone.  In (z_1, z_2) it enters as [Su -Su; -Su Su], and when the cone is inactive (t -> 0) Su
         * grows like 1/mu, so eliminating z_1 before z_2 cancels K_2 + Su - Su (K_1 + Su)^{-1} Su. */
        for (int k = 0; k < n; ++k) {
            for (int r = 0; r < NB; ++r) F.M[r * NB + 3 * n + k] += F.M[r * NB + 2 * n + k];
            for (int cc = 0; cc < NB; ++cc) F.M[(3 * n + k) * NB + cc] += F.M[(2 * n + k) * NB + cc];
        }
        for (int k = 0; k < n; ++k)
            for (int l = 0; l < n; ++l) F.M[(2 * n + k) * NB + 2 * n + l] += F.soc_Su[k * n + l];
        if (bad || chol(NB, F.M, NB)) { status = -6; break; }

        /* ---- affine direction: kappa = 0 ---- */
        memset(P.k1, 0, sizeof(double) * R); memset(P.k2, 0, sizeof(double) * R);
        P.kyv[0] = P.kyv[1] = 0; for (int k = 0; k < q; ++k) P.ksoc[k] = 0;
        for (int e = 0; e < d; ++e) { block_t *B = &P.blk[e]; memset(B->k3, 0, sizeof(double) * R); memset(B->k4, 0, sizeof(double) * R); B->k5 = B->k6 = 0; }
        double db[MAXNB], dnu[2][MAXNW];
        double dsyv[2], dssoc[MAXN + 1], dlsoc[MAXN + 1], ds56[2];
        double amax = 1e300, c1 = 0, c2 = 0; /* sums: s.dl + l.ds ; ds.dl */
        double dlyv[2], sm;
        /* (the first iteration of a warm solve has no predictor: kappa = mu_ref / s, no second-order term) */
        if (first_warm) sm = mu_ref;
        else {
        lagr_grad(&P, gb);
        newton_solve(&P, &F, gb, rp, db, dnu);
        slack_dir(&P, db, ds1, ds2, dsyv, dssoc, -1, NULL, NULL, NULL, NULL);
        for (int r = 0; r < R; ++r) { ROW(P.s1[r], P.l1[r], ds1[r], 0.0, dl1[r]); ROW(P.s2[r], P.l2[r], ds2[r], 0.0, dl2[r]); }
        ROW(P.syv[0], P.lyv[0], dsyv[0], 0.0, dlyv[0]); ROW(P.syv[1], P.lyv[1], dsyv[1], 0.0, dlyv[1]);
        for (int i = 0; i < q; ++i) {
            double a = -P.lsoc[i];
            for (int k = 0; k < q; ++k) a -= W2[i * ldq + k] * dssoc[k];
            dlsoc[i] = a;
        }
        amax = fmin(amax, fmin(soc_max_step(q, P.ssoc, dssoc), soc_max_step(q, P.lsoc, dlsoc)));
        for (int k = 0; k < q; ++k) { c1 += P.ssoc[k] * dlsoc[k] + P.lsoc[k] * dssoc[k]; c2 += dssoc[k] * dlsoc[k]; }
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            double *ds3 = ds1 + 4 * R + (size_t)e * 4 * R, *ds4 = ds3 + R, *dl3 = ds3 + 2 * R, *dl4 = ds3 + 3 * R;
            slack_dir(&P, db, NULL, NULL, NULL, NULL, e, B->dw, ds3, ds4, ds56);
            double dl5, dl6;
            for (int r = 0; r < R; ++r) { ROW(B->s3[r], B->l3[r], ds3[r], 0.0, dl3[r]); ROW(B->s4[r], B->l4[r], ds4[r], 0.0, dl4[r]); }
            ROW(B->s5, B->l5, ds56[0], 0.0, dl5); ROW(B->s6, B->l6, ds56[1], 0.0, dl6);
            /* corrector multipliers need ds*dl of the affine step: store in k3/k4 (as products) */
            for (int r = 0; r < R; ++r) { B->k3[r] = ds3[r] * dl3[r]; B->k4[r] = ds4[r] * dl4[r]; }
            B->k5 = ds56[0] * dl5; B->k6 = ds56[1] * dl6;
            memcpy(B->dwa, B->dw, sizeof(double) * NW);
        }
        for (int r = 0; r < R; ++r) { P.k1[r] = ds1[r] * dl1[r]; P.k2[r] = ds2[r] * dl2[r]; }
        P.kyv[0] = dsyv[0] * dlyv[0]; P.kyv[1] = dsyv[1] * dlyv[1];
        const double al_aff = fmin(1.0, amax);
        double mu_aff = (gap + al_aff * c1 + al_aff * al_aff * c2) / deg;
        double sig = mu_aff / mu; sig = sig < 0 ? 0 : (sig > 1 ? 1 : sig); sig = sig * sig * sig;
        sm = sig * mu;
        }
        /* kappa = (sigma mu - ds_a dl_a) / s */
        for (int r = 0; r < R; ++r) { P.k1[r] = (sm - P.k1[r]) / P.s1[r]; P.k2[r] = (sm - P.k2[r]) / P.s2[r]; }
        P.kyv[0] = (sm - P.kyv[0]) / P.syv[0]; P.kyv[1] = (sm - P.kyv[1]) / P.syv[1];
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            for (int r = 0; r < R; ++r) { B->k3[r] = (sm - B->k3[r]) / B->s3[r]; B->k4[r] = (sm - B->k4[r]) / B->s4[r]; }
            B->k5 = (sm - B->k5) / B->s5; B->k6 = (sm - B->k6) / B->s6;
        }
        {   /* cone: kappa = sigma mu s^{-1} - W^{-1} (lt \ ((W^{-1} ds_a) o (W dl_a))) */
            double a1[MAXN + 1], a2[MAXN + 1], pr[MAXN + 1], qv[MAXN + 1];
            for (int i = 0; i < q; ++i) {
                double u1 = 0, u2 = 0;
                for (int k = 0; k < q && !first_warm; ++k) { u1 += F.Wsoci[i * ldq + k] * dssoc[k]; u2 += F.Wsoc[i * ldq + k] * dlsoc[k]; }
                a1[i] = u1; a2[i] = u2;
            }
            soc_prod(q, a1, a2, pr);
            soc_div(q, lt, pr, qv);
            double dets = soc_det(q, P.ssoc);
            for (int i = 0; i < q; ++i) {
                double a = 0;
                for (int k = 0; k < q; ++k) a += F.Wsoci[i * ldq + k] * qv[k];
                P.ksoc[i] = sm * (i == 0 ? P.ssoc[0] : -P.ssoc[i]) / dets - (first_warm ? 0.0 : a);
            }
        }
        lagr_grad(&P, gb);
        newton_solve(&P, &F, gb, rp, db, dnu);
        amax = 1e300; c1 = c2 = 0;
        slack_dir(&P, db, ds1, ds2, dsyv, dssoc, -1, NULL, NULL, NULL, NULL);
        for (int r = 0; r < R; ++r) { ROW(P.s1[r], P.l1[r], ds1[r], P.k1[r], dl1[r]); ROW(P.s2[r], P.l2[r], ds2[r], P.k2[r], dl2[r]); }
        ROW(P.syv[0], P.lyv[0], dsyv[0], P.kyv[0], dlyv[0]); ROW(P.syv[1], P.lyv[1], dsyv[1], P.kyv[1], dlyv[1]);
        for (int i = 0; i < q; ++i) {
            double a = P.ksoc[i] - P.lsoc[i];
            for (int k = 0; k < q; ++k) a -= W2[i * ldq + k] * dssoc[k];
            dlsoc[i] = a;
        }
        amax = fmin(amax, fmin(soc_max_step(q, P.ssoc, dssoc), soc_max_step(q, P.lsoc, dlsoc)));
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            double *ds3 = ds1 + 4 * R + (size_t)e * 4 * R, *ds4 = ds3 + R, *dl3 = ds3 + 2 * R, *dl4 = ds3 + 3 * R;
            slack_dir(&P, db, NULL, NULL, NULL, NULL, e, B->dw, ds3, ds4, ds56);
            double dl5, dl6;
            for (int r = 0; r < R; ++r) { ROW(B->s3[r], B->l3[r], ds3[r], B->k3[r], dl3[r]); ROW(B->s4[r], B->l4[r], ds4[r], B->k4[r], dl4[r]); }
            ROW(B->s5, B->l5, ds56[0], B->k5, dl5); ROW(B->s6, B->l6, ds56[1], B->k6, dl6);
            B->k5 = dl5; B->k6 = dl6; /* reuse as storage of the dual step */
        }
        double al = fmin(1.0, 0.99 * amax);
        if (getenv("GCS_ORACLE_DEBUG3") && (!getenv("GCS_ORACLE_DEBUG_V") || atoi(getenv("GCS_ORACLE_DEBUG_V")) == dbg_vertex)) fprintf(stderr, "  it %d mu %.3e rd %.2e rp %.2e sigma*mu %.2e amax %.4e yv %.4e t %.4e\n", it, mu, rdmax, rpmax, sm, amax, P.beta[4*n], P.beta[4*n+1]);
        for (int tries = 0; tries < 40; ++tries) {   /* keep both cone points strictly inside despite round-off */
            double s2[MAXN + 1] = {0}, l2[MAXN + 1] = {0};
            for (int k = 0; k < q; ++k) { s2[k] = P.ssoc[k] + al * dssoc[k]; l2[k] = P.lsoc[k] + al * dlsoc[k]; }
            if (soc_interior(q, s2) && soc_interior(q, l2)) break;
            al *= 0.7;
            if (getenv("GCS_ORACLE_DEBUG4")) fprintf(stderr, "[oracle] cone guard backtrack it=%d\n", it);
        }
        stalled = al < 1e-3;
        for (int k = 0; k < NB; ++k) P.beta[k] += al * db[k];
        for (int s = 0; s < 2; ++s) for (int k = 0; k < NW; ++k) P.nu[s][k] += al * dnu[s][k];
        for (int r = 0; r < R; ++r) { P.l1[r] += al * dl1[r]; P.l2[r] += al * dl2[r]; }
        P.lyv[0] += al * dlyv[0]; P.lyv[1] += al * dlyv[1];
        for (int k = 0; k < q; ++k) P.lsoc[k] += al * dlsoc[k];
        for (int e = 0; e < d; ++e) {
            block_t *B = &P.blk[e];
            double *ds3 = ds1 + 4 * R + (size_t)e * 4 * R, *dl3 = ds3 + 2 * R, *dl4 = ds3 + 3 * R;
            for (int k = 0; k < NX; ++k) B->O[k] += al * B->dw[k];
            B->y += al * B->dw[2 * n];
            for (int r = 0; r < R; ++r) { B->l3[r] += al * dl3[r]; B->l4[r] += al * dl4[r]; }
            B->l5 += al * B->k5; B->l6 += al * B->k6;
        }
    }
    if (status != 0 && getenv("GCS_ORACLE_DEBUG")) fprintf(stderr, "[oracle] vertex solve status %d after %d iterations (d=%d m=%d warm=%d)\n", status, it, d, m, use_warm);
    it_total += it;
    if (status != 0 && warm) warm[0] = 0.0;                       /* no restart from a solve that failed */
    if (status != 0 && use_warm) { use_warm = 0; dbg_kind = 3; goto restart; }  /* a failed warm solve is repeated cold */
    if (warm && ws_adapt && status == 0) {
        /* the far-warm threshold of this vertex learns from its own solves */
        const double theta = fmax(WS_NEAR, warm[2] > 0 ? warm[2] : WS_COLD_DT);
        if (!use_warm) {
            warm[3] = (double)it;                                                        /* iterations of a cold solve */
            if (was_warm) warm[2] = fmax(WS_NEAR, WS_SHRINK * dT);                       /* the warm attempt failed */
            else if (dT >= 0) warm[2] = fmin(WS_THETA_MAX, WS_GROW * theta);             /* cold because the targets moved: try further next time */
        } else if (dT > WS_NEAR && warm[3] > 0 && (double)it > warm[3]) warm[2] = fmax(WS_NEAR, WS_SHRINK * dT);   /* dearer than cold */
    }
    it = it_total;
    /* un-centre and report */
    const double yvv = P.beta[4 * n];
    for (int k = 0; k < n; ++k) {
        xv[k] = P.beta[k] + cen[k]; xv[n + k] = P.beta[n + k] + cen[k];
        zv[k] = P.beta[2 * n + k] + yvv * cen[k]; zv[n + k] = P.beta[3 * n + k] + yvv * cen[k];
    }
    *yv = yvv;
    for (int e = 0; e < d; ++e) {
        block_t *B = &P.blk[e];
        for (int k = 0; k < n; ++k) {
            double o1 = B->O[k] + B->y * cen[k], o2 = B->O[n + k] + B->y * cen[k];
            if (B->out) { CW(k, e) = o1; CW(n + k, e) = o2; }
            else { CW(k, e) = TW(k, e); CW(n + k, e) = o1; }
        }
        CW(2 * n, e) = B->y;
    }
    (void)c;
    return status == 0 ? it : -(100 + it);
#undef TW
#undef CW
#undef KB
#undef ROW
}

/* ------------------------------------------------------------------ vertex step over the whole graph
 * (admm_solver_v3.py:469-540).  targets = zedge - mu_scale * mu.  Returns the number of
 * sub-problems whose inner solver did not converge; ipm_iters_total accumulates iterations. */
/* diagnostics: when set, oracle_vertex_step records the Newton iteration count of every vertex (-1 = failed) */
static int *g_iters_out = 0, *g_kind_out = 0; static double *g_dt_out = 0, *g_rd_out = 0;
void oracle_set_rd_out(double *buf) { g_rd_out = buf; }      /* diagnostic (tools/warm_accuracy_study.py): dual residual at the stop of each vertex's solve */
void oracle_set_dt_out(double *buf) { g_dt_out = buf; }
void oracle_set_iters_out(int *buf) { g_iters_out = buf; }
void oracle_set_kind_out(int *buf) { g_kind_out = buf; }

int oracle_vertex_step(const oracle_graph *G, const double *zedge, const double *mu, double mu_scale,
                       double rho, const oracle_inner_params *ip, double *copy,
                       double *xv, double *zv, double *yv, long *ipm_iters_total, int nthreads)
{
    const int n = G->n, c = 2 * n + 1, NI = G->NI > 0 ? G->NI : 2 * G->E, E = G->E;
    int fails = 0; long iters = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : fails, iters)
#endif
    for (int v = 0; v < G->V; ++v) {
        const int lo = G->inc_ptr[v], d = G->inc_ptr[v + 1] - lo;
        int d_in = 0;
        for (int k = 0; k < d; ++k) d_in += !G->inc_out[lo + k];
        double Tst[2 * MAXNW * 64];
        double *T = (d <= 64) ? Tst : (double *)malloc(sizeof(double) * 2 * c * d), *C = T + c * (d > 0 ? d : 1);
        for (int k = 0; k < d; ++k) {
            const int e = G->inc_edge[lo + k];
            for (int w = 0; w < c; ++w) T[w * d + k] = zedge[w * E + e] - mu_scale * mu[w * NI + lo + k];
        }
        const int p0 = G->poly_ptr[v], m = G->poly_ptr[v + 1] - p0;
        dbg_vertex = v;
        double xv_t[2 * MAXN], zv_t[2 * MAXN], yv_t = 0.0;
        int r = oracle_solve_vertex(n, m, G->poly_A + (size_t)p0 * n, G->poly_b + p0, G->center + (size_t)v * n,
                                    d, d_in, v == G->src, v == G->dst, T, rho, ip, C, xv_t, zv_t, &yv_t,
                                    ip->warm ? ip->warm + ip->warm_ptr[v] : NULL);
        if (r < 0) fails += 1; else iters += r;
        if (g_iters_out) g_iters_out[v] = r;
        if (g_kind_out) g_kind_out[v] = dbg_kind;
        if (g_dt_out) g_dt_out[v] = dbg_dt;
        if (g_rd_out) g_rd_out[v] = dbg_rd;
        /* a failed inner solve keeps the vertex's previous copy columns and outputs (the reference's intent at
         * admm_solver_v3.py:524-538; its own branch would raise) and is counted */
        if (r >= 0) {
            for (int k = 0; k < 2 * n; ++k) { xv[(size_t)v * 2 * n + k] = xv_t[k]; zv[(size_t)v * 2 * n + k] = zv_t[k]; }
            yv[v] = yv_t;
            for (int k = 0; k < d; ++k)
                for (int w = 0; w < c; ++w) copy[w * NI + lo + k] = C[w * d + k];
        }
        if (T != Tst) free(T);
    }
    if (ipm_iters_total) *ipm_iters_total += iters;
    return fails;
}

/* ------------------------------------------------------------------ edge step + dual update + the five sums
 * (admm_solver_v3.py:543-614).  sums = [ |r|^2, |dz|^2, |copy|^2, |zedge|^2, |mu|^2 ] */
void oracle_edge_step(const oracle_graph *G, const double *copy, double *zedge, double *mu,
                      double mu_scale, double sums[5])
{
    const int n = G->n, c = 2 * n + 1, NI = G->NI > 0 ? G->NI : 2 * G->E, E = G->E;
    double s_r = 0, s_dz = 0, s_ax = 0, s_bz = 0, s_mu = 0;
    for (int e = 0; e < E; ++e) {
        const int it = G->edge_inc_tail[e], ih = G->edge_inc_head[e];
        for (int w = 0; w < c; ++w) {
            const double cu = copy[w * NI + it], cw = copy[w * NI + ih];
            const double zn = 0.5 * (cu + cw), zo = zedge[w * E + e];
            const double ru = cu - zn, rw = cw - zn;
            const double mu_u = mu_scale * mu[w * NI + it] + ru, mu_w = mu_scale * mu[w * NI + ih] + rw;
            mu[w * NI + it] = mu_u; mu[w * NI + ih] = mu_w;
            zedge[w * E + e] = zn;
            const double we = G->edge_counted ? (double)G->edge_counted[e] : 1.0;
            const double wt = G->inc_counted ? (double)G->inc_counted[it] : 1.0, wh = G->inc_counted ? (double)G->inc_counted[ih] : 1.0;
            s_r += wt * ru * ru + wh * rw * rw;
            s_dz += we * (zn - zo) * (zn - zo);
            s_ax += wt * cu * cu + wh * cw * cw;
            s_bz += we * zn * zn;
            s_mu += wt * mu_u * mu_u + wh * mu_w * mu_w;
        }
    }
    sums[0] = s_r; sums[1] = s_dz; sums[2] = s_ax; sums[3] = s_bz; sums[4] = s_mu;
}

typedef struct {
    double rho, tau_incr, tau_decr, nu;
    int it_rho_limit;        /* rho adapts only while it < this (frac*MAX_IT = 100) */
    double eps_abs, eps_rel;
    int max_it;
} oracle_admm_params;

/* ------------------------------------------------------------------ the loop (admm_solver_v3.py:621-733)
 * trace: per iteration 6 doubles (rho after adaptation, pri, dual, eps_pri, eps_dual, inner failures).
 * Returns the iteration count `it` at exit as the reference reports it. */
/* the loop from iteration it_start (ap->rho = the penalty in force there; ap->max_it = the last iteration that may run; the trace
 * row of iteration it is trace + (it - it_start) * 6): a run can be continued by a second call -- bench.py's cpu_baseline times the
 * same iteration window as the GPU after an untimed advance */
int oracle_admm_run_from(const oracle_graph *G, const oracle_admm_params *ap, const oracle_inner_params *ip,
                         double *zedge, double *mu, double *copy, double *xv, double *zv, double *yv,
                         double *trace, int *status_out, long *ipm_iters_total, int nthreads, int it_start, double *rho_out)
{
    const int n = G->n;
    const double nx = G->nx_global > 0 ? G->nx_global : (4.0 * n + 1) * (G->V + 2.0 * G->E);
    const double nmu = G->nmu_global > 0 ? G->nmu_global : (4.0 * n + 2) * G->E;
    double rho = ap->rho, mu_scale = 1.0;
    int it = it_start, status = 1; /* 1 = max_it reached, 0 = converged, 2 = diverged */
    while (it <= ap->max_it) {
        int fails = oracle_vertex_step(G, zedge, mu, mu_scale, rho, ip, copy, xv, zv, yv, ipm_iters_total, nthreads);
        double s[5];
        oracle_edge_step(G, copy, zedge, mu, mu_scale, s);
        mu_scale = 1.0;
        if (!isfinite(s[0] + s[1] + s[2] + s[3] + s[4])) { status = 2; break; }
        const double pri = sqrt(s[0]), dual = rho * sqrt(2.0 * s[1]);
        if (pri >= ap->nu * dual && it < ap->it_rho_limit) { rho *= ap->tau_incr; mu_scale = 1.0 / ap->tau_incr; }
        else if (dual >= ap->nu * pri && it < ap->it_rho_limit) { rho *= 1.0 / ap->tau_decr; mu_scale = ap->tau_incr; }
        const double eps_pri = sqrt(nx) * ap->eps_abs + ap->eps_rel * fmax(sqrt(s[2]), sqrt(2.0 * s[3]));
        const double eps_dual = sqrt(nmu) * ap->eps_abs + ap->eps_rel * mu_scale * sqrt(s[4]);
        double *tr = trace + (size_t)(it - it_start) * 6;
        tr[0] = rho; tr[1] = pri; tr[2] = dual; tr[3] = eps_pri; tr[4] = eps_dual; tr[5] = fails;
        if (pri < eps_pri && dual < eps_dual) { status = 0; break; }
        it += 1;
    }
    /* a pending rescale of mu (set on the last executed iteration) is applied so the state is self-consistent */
    if (mu_scale != 1.0) { const size_t N = (size_t)(2 * n + 1) * (G->NI > 0 ? G->NI : 2 * G->E); for (size_t i = 0; i < N; ++i) mu[i] *= mu_scale; }
    *status_out = status;
    if (rho_out) *rho_out = rho;
    return it;
}

int oracle_admm_run(const oracle_graph *G, const oracle_admm_params *ap, const oracle_inner_params *ip,
                    double *zedge, double *mu, double *copy, double *xv, double *zv, double *yv,
                    double *trace, int *status_out, long *ipm_iters_total, int nthreads)
{
    return oracle_admm_run_from(G, ap, ip, zedge, mu, copy, xv, zv, yv, trace, status_out, ipm_iters_total, nthreads, 1, NULL);
}

/* GCS_utils.py:184-211 on the last iterate */
double oracle_compute_cost(const oracle_graph *G, const double *zv, const double *zedge, double eps_edge)
{
    const int n = G->n;
    double len = 0, pen = 0;
    for (int v = 0; v < G->V; ++v) {
        double s = 0;
        for (int k = 0; k < n; ++k) { double dlt = zv[(size_t)v * 2 * n + k] - zv[(size_t)v * 2 * n + n + k]; s += dlt * dlt; }
        len += sqrt(s);
    }
    for (int e = 0; e < G->E; ++e) pen += eps_edge * (G->edge_counted ? (double)G->edge_counted[e] : 1.0) * zedge[(size_t)(2 * n) * G->E + e];
    return len + pen;
}

/* one pass of the loop control (admm_solver_v3.py:697-733) on globally reduced sums; state[] = {rho, mu_scale, it, status}
 * (status: -1 running, 0 converged, 1 max_it, 2 diverged); writes the 6-double trace record. */
void oracle_control(const oracle_graph *G, const oracle_admm_params *ap, const double s[5], double state[4], double fails, double *tr)
{
    const int n = G->n;
    const double nx = G->nx_global > 0 ? G->nx_global : (4.0 * n + 1) * (G->V + 2.0 * G->E);
    const double nmu = G->nmu_global > 0 ? G->nmu_global : (4.0 * n + 2) * G->E;
    double rho = state[0], mu_scale = 1.0;
    const int it = (int)state[2];
    if (state[3] != -1.0) return;
    if (!isfinite(s[0] + s[1] + s[2] + s[3] + s[4])) { state[3] = 2; return; }
    const double pri = sqrt(s[0]), dual = rho * sqrt(2.0 * s[1]);
    if (pri >= ap->nu * dual && it < ap->it_rho_limit) { rho *= ap->tau_incr; mu_scale = 1.0 / ap->tau_incr; }
    else if (dual >= ap->nu * pri && it < ap->it_rho_limit) { rho *= 1.0 / ap->tau_decr; mu_scale = ap->tau_incr; }
    const double eps_pri = sqrt(nx) * ap->eps_abs + ap->eps_rel * fmax(sqrt(s[2]), sqrt(2.0 * s[3]));
    const double eps_dual = sqrt(nmu) * ap->eps_abs + ap->eps_rel * mu_scale * sqrt(s[4]);
    state[0] = rho; state[1] = mu_scale;
    if (tr) { tr[0] = rho; tr[1] = pri; tr[2] = dual; tr[3] = eps_pri; tr[4] = eps_dual; tr[5] = fails; }
    if (pri < eps_pri && dual < eps_dual) { state[3] = 0; return; }
    state[2] = it + 1;
    if (it + 1 > ap->max_it) state[3] = 1;
}
