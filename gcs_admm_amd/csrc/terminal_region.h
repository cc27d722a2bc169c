// terminal_region.h -- x-update of a terminal ('s' / 't') that is a REGION, one workgroup per terminal.
//
// The reference builds its terminals as points (utils.py:12-28: boxes of half-width 1e-6; closed form here, special_vertex.h), but its
// vertex update constrains them like any set (admm_solver_v3.py:415-440) with delta_sv / delta_tv in the flow rows (:450-464).  For
// v = 's' (v = 't' mirrored: live side = incoming, only [O_e]_1 penalised):
//   (6) with y_v <= 1 forces y_v = 1 and y_e = 0 on the incoming side, hence O_e = 0 there (rows 3 of a bounded set);
//   (7) reads z_v = x_v = sum_{e out} O_e, (6) sum_{e out} y_e = 1;
//   rows 1 (A x_i <= b), rows 2 (0 <= 0) and rows 4 (A (x_i - O_{e,i}) <= (1 - y_e) b) are sums of rows 3 of the live blocks and of
//   the two equalities: redundant; 0 <= y_e <= 1 follows from rows 3 and sum y = 1.
// Left, over the live blocks p_e = (O_e in R^{2n}, y_e) and t:
//   min  t + sum_e [ eps y_e + rho/2 ( |[O_e]_1 - T1_e|^2 + out_e |[O_e]_2 - T2_e|^2 + (y_e - Ty_e)^2 ) ]
//   s.t. A [O_e]_i <= y_e b (i = 1, 2),   sum_e y_e = 1,   | sum_e ([O_e]_1 - [O_e]_2) | <= t.
// Same primal-dual method as the generic vertex programs (Mehrotra predictor-corrector, one step length, NT scaling of the cone with t
// eliminated in closed form, sigma = (mu_aff / mu)^3, stop on mu), cold from y_e = 1 / L, O_e = y_e (c, c), or warm from the record the terminal's
// previous solve left (below: the rule of warm_start.h with a fixed threshold; 7.6 -> ~3.8 Newton iterations per solve).  The Hessian is block
// diagonal (2n+1 per live block) plus the cone and the equality, which enter through F = [I, -I, 0; 0, 0, 1]: the blocks are eliminated
// onto (du, dnu), n + 1 unknowns.  The CPU restatement is oracle/gcs_oracle.c solve_terminal_region (checked there against the
// sub-problem as written, tests/test_terminal_region.py).
//
// Written against an executor EX (tid, nthreads, sync, reduce3; stamp: a no-op outside the timing build) so that the same body runs as one workgroup on the device
// (terminal_region.hip) and serially on the host (tests/hostemu/term_emu.cpp).  The work arrays (terminal_ws_doubles) are given by the
// caller: LDS where they fit (a terminal with 6 live edges and 5 facets in R^2: 7 KB), the handle's HBM workspace otherwise (degree 40 in
// R^6: 200 KB).  The solve is a chain of short dependent phases: it is bound by the latency of its storage and of its barriers, which is
// why small terminals run in ONE wavefront (barriers and reductions without LDS round trips).
#pragma once
#include <stdint.h>

#include "gcs_math.h"

namespace gcs_term {

using gcs_math::rcp;
using gcs_math::sqrt_nr;
// a / b and sqrt in the serial (thread 0) parts: refined hardware estimates on the device (gcs_math.h), IEEE on the host
GCS_HD double fdiv(double a, double b) { return a * rcp(b); }

constexpr double TERM_REG = 1e-7;         // Tikhonov term on every unknown, as in the vertex programs (REG_DELTA)
constexpr double TERM_CHOL_SKIP = 1e-12;  // pivot floor relative to the diagonal entry, as in the vertex programs
// warm start: the rule and the constants of the vertex programs (warm_start.h) with a fixed threshold.  Record of a terminal, in doubles:
//   [0] valid [1] rho [2] live blocks [3] nu | per live block: p (2n+1) | targets (2n+1) | row duals (2m)
// (t and the cone's dual are re-centred at the restart: t^2 - mu_ref t - |u|^2 = 0, lambda = (1, -u / t); not kept)
constexpr double TERM_WS_KAPPA = 3e-3, TERM_WS_MU_MIN = 1e-7, TERM_WS_COLD_DT = 1.0, TERM_WS_SAVE = 10.0, TERM_WS_COLD_REF = 1e-4;
constexpr int TERM_W_HDR = 4;
inline long long terminal_record_doubles(int n, int m, int L) { return TERM_W_HDR + (long long)L * (2 * (2 * n + 1) + 2 * m); }

inline long long terminal_ws_doubles(int n, int m, int L)
{
    const long long NW = 2 * n + 1, R = 2 * m, NF = n + 1;
    return m + (long long)m * n + (long long)L * (6 * NW + 5 * R + NW * NW + NW * NF + NF);
}

template <int N> struct TermShared {
    static constexpr int Q = N + 1, NF = N + 1;
    double ssoc[Q], lsoc[Q], ksoc[Q], dssoc[Q], dlsoc[Q], lt[Q];
    double Wsoc[Q * Q], Wsoci[Q * Q], W2[Q * Q], Su[N * N], cv[N];
    double S[NF * NF], z[NF], su[NF];
    double M[NF * NF], x[NF];         // the small system of a Newton solve (thread 0; dynamic row swaps: kept out of registers)
    double c0, t, nu, dt, dnu, mu, gap, sm, al, mu_ref;
    int status, stalled, stop, use_warm, saved, save_now;
};

// everything the solve reads and writes besides its workspace
template <class T> struct TermProblem {
    int m, d, d_in, is_src;
    const double *A, *bc, *cen;         // facets [m][N], centred right-hand sides b - A c, centre
    const int *inc_edge;                // [d] edge ids of the incidences of this vertex
    int inc_lo;                         // first incidence (incidence-major column of local incidence k: inc_lo + k)
    int E, NI, edge_major;
    const T *zedge, *mu;
    T *copy;
    double *xv, *zv, *yv;               // rows of this vertex
    double rho, mu_scale, eps_edge, ipm_tol;
    int ipm_max_iter;
    double *warm;                       // the terminal's record (terminal_record_doubles, zero = none yet); nullptr: every solve starts cold
};

template <int N, class T>
GCS_HD int state_column(const TermProblem<T> &P, int k)
{
    return P.edge_major ? P.inc_edge[k] + (k >= P.d_in ? 0 : P.E) : P.inc_lo + k;
}
template <int N, class T>
GCS_HD double target(const TermProblem<T> &P, int word, int k)
{
    return (double)P.zedge[(size_t)word * P.E + P.inc_edge[k]] - P.mu_scale * (double)P.mu[(size_t)word * P.NI + state_column<N, T>(P, k)];
}

template <int Q> GCS_HD void soc_prod(const double *a, const double *b, double *o)
{
    double d = 0;
    for (int k = 0; k < Q; ++k) d += a[k] * b[k];
    o[0] = d;
    for (int k = 1; k < Q; ++k) o[k] = a[0] * b[k] + b[0] * a[k];
}
template <int Q> GCS_HD void soc_div(const double *l, const double *d, double *x)
{   // l o x = d
    double ld1 = 0;
    for (int k = 1; k < Q; ++k) ld1 += l[k] * d[k];
    x[0] = fdiv(l[0] * d[0] - ld1, gcs_math::soc_det<Q>(l));
    const double il0 = rcp(l[0]);
    for (int k = 1; k < Q; ++k) x[k] = (d[k] - x[0] * l[k]) * il0;
}

// Returns the iteration count (>= 0) or a negative status; every thread of the workgroup returns the same value.
template <int N, class T, class EX>
GCS_HD int terminal_region_solve(EX &ex, const TermProblem<T> &P, double *ws, TermShared<N> &sh)
{
    constexpr int NW = 2 * N + 1, NF = N + 1, Q = N + 1;
    const int m = P.m, R = 2 * m;
    const int lo = P.is_src ? P.d_in : 0, hi = P.is_src ? P.d : P.d_in, L = hi - lo;
    const int tid = ex.tid(), nt = ex.nthreads();
    if (L <= 0) return -2;
    double *b = ws, *Al = b + m, *pp = Al + m * N, *tg = pp + L * NW, *qd = tg + L * NW, *rhs = qd + L * NW, *hr = rhs + L * NW, *dp = hr + L * NW;
    double *sl = dp + L * NW, *lam = sl + L * R, *kap = lam + L * R, *ds = kap + L * R, *dl = ds + L * R;
    double *H = dl + L * R, *X = H + (size_t)L * NW * NW, *zc = X + (size_t)L * NW * NF;
    const double *A = Al;
    // ---- prologue: the facets next to the other work arrays (they are read in every phase: LDS, not HBM, where the arrays are in LDS),
    //      raw right-hand sides, targets, the start y_e = 1 / L, O_e = y_e (c, c), t = 1
    for (int j = tid; j < m; j += nt) {
        double a = P.bc[j];
        for (int k = 0; k < N; ++k) { const double ajk = P.A[j * N + k]; Al[j * N + k] = ajk; a += ajk * P.cen[k]; }
        b[j] = a;
    }
    const double invL = 1.0 / L;
    const int W_PER = 2 * NW + R;
    double *const rec = P.warm;
    const bool comparable = rec != nullptr && rec[0] == 1.0 && rec[1] == P.rho && (int)rec[2] == L;
    double moved = 0.0;         // largest |target - target of the record| over the penalised words
    for (int idx = tid; idx < L * NW; idx += nt) {
        const int e = idx / NW, k = idx - e * NW, ge = lo + e;
        double tgt, q;
        if (k < N) { tgt = target<N, T>(P, P.is_src ? k : N + k, ge); q = P.rho; }
        else if (k < 2 * N) { tgt = P.is_src ? target<N, T>(P, k, ge) : 0.0; q = P.is_src ? P.rho : 0.0; }
        else { tgt = target<N, T>(P, 2 * N, ge); q = P.rho; }
        tg[idx] = tgt; qd[idx] = q;
        if (comparable && q > 0) moved = fmax(moved, fabs(tgt - rec[TERM_W_HDR + e * W_PER + NW + k]));
    }
    {
        double neg = -moved, u1 = 0.0, u2 = 0.0;
        ex.reduce3(neg, u1, u2);
        const double dT = -neg * P.rho;
        if (tid == 0) {
            sh.use_warm = comparable && dT <= TERM_WS_COLD_DT;
            sh.mu_ref = sh.use_warm ? fmax(TERM_WS_MU_MIN, TERM_WS_KAPPA * dT) : TERM_WS_COLD_REF;
        }
    }
    ex.sync();
    const int deg = L * R + 1;
    int it = 0, it_total = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {          // a warm solve that fails is repeated cold
    const bool use_warm = sh.use_warm != 0;
    // ---- the start: the record (cone pair re-centred at mu_ref), or y_e = 1 / L, O_e = y_e (c, c), t = 1
    if (use_warm) {
        for (int idx = tid; idx < L * NW; idx += nt) { const int e = idx / NW, k = idx - e * NW; pp[idx] = rec[TERM_W_HDR + e * W_PER + k]; }
        for (int idx = tid; idx < L * R; idx += nt) { const int e = idx / R, r = idx - e * R; lam[idx] = rec[TERM_W_HDR + e * W_PER + 2 * NW + r]; }
    } else {
        for (int idx = tid; idx < L * NW; idx += nt) { const int k = idx % NW; pp[idx] = k < 2 * N ? P.cen[k < N ? k : k - N] * invL : invL; }
    }
    ex.sync();
    if (tid == 0) {
        sh.status = -1; sh.stalled = 0; sh.stop = 0; sh.saved = 0; sh.save_now = 0;
        for (int k = 0; k < Q; ++k) sh.lsoc[k] = 0.0;
        if (use_warm) {
            double u[N], uu = 0;
            for (int k = 0; k < N; ++k) { double a = 0; for (int e = 0; e < L; ++e) a += pp[e * NW + k] - pp[e * NW + N + k]; u[k] = a; uu += a * a; }
            sh.t = 0.5 * (sh.mu_ref + sqrt(sh.mu_ref * sh.mu_ref + 4.0 * uu));
            sh.lsoc[0] = 1.0;
            for (int k = 0; k < N; ++k) sh.lsoc[1 + k] = -u[k] / sh.t;
            sh.nu = rec[3];
        } else { sh.t = 1.0; sh.nu = 0.0; sh.mu_ref = TERM_WS_COLD_REF; }
    }
    ex.sync();
    // one Newton solve with the multipliers kap (rows) / sh.ksoc (cone) in place of the duals: leaves dp, ds, sh.dssoc, sh.dt, sh.dnu
    auto newton = [&]() {
        const double gt = 1.0 - sh.ksoc[0], ct = fdiv(gt, sh.c0);
        // right-hand sides, one entry per thread (sums in registers): -(grad f + G' kap) of the block, the cone's part, t eliminated
        for (int idx = tid; idx < L * NW; idx += nt) {
            const int e = idx / NW, k = idx - e * NW;
            const double *ke = kap + e * R;
            double a = -(qd[idx] * (pp[idx] - tg[idx]) + TERM_REG * pp[idx]);
            if (k < 2 * N) {
                const int i = k < N ? 0 : 1, kk = k - i * N;
                for (int j = 0; j < m; ++j) a -= A[j * N + kk] * ke[i * m + j];
                const double c = sh.ksoc[1 + kk] + sh.cv[kk] * ct;
                a += i == 0 ? c : -c;
            } else {
                double gy = P.eps_edge + sh.nu;
                for (int r = 0; r < R; ++r) gy -= b[r < m ? r : r - m] * ke[r];
                a -= gy;
            }
            rhs[idx] = a;
        }
        ex.sync();
        ex.stamp(5);       // right-hand sides
        for (int idx = tid; idx < L * NF; idx += nt) {      // the block's part of z = sum_e X_e' rhs_e
            const int e = idx / NF, c = idx - e * NF;
            const double *Xe = X + (size_t)e * NW * NF, *r = rhs + e * NW;
            double a = 0;
            for (int k = 0; k < NW; ++k) a += Xe[k * NF + c] * r[k];
            zc[idx] = a;
        }
        for (int e = tid; e < L; e += nt) {                 // H_e^{-1} rhs_e
            const double *Le = H + (size_t)e * NW * NW, *r = rhs + e * NW;
            double *h = hr + e * NW;
            for (int i = 0; i < NW; ++i) { double a = r[i]; for (int k = 0; k < i; ++k) a -= Le[i * NW + k] * h[k]; h[i] = a * Le[i * NW + i]; }
            for (int i = NW - 1; i >= 0; --i) { double a = h[i]; for (int k = i + 1; k < NW; ++k) a -= Le[k * NW + i] * h[k]; h[i] = a * Le[i * NW + i]; }
        }
        ex.sync();
        ex.stamp(6);       // X' rhs, H^{-1} rhs
        for (int c = tid; c < NF; c += nt) { double a = 0; for (int e = 0; e < L; ++e) a += zc[e * NF + c]; sh.z[c] = a; }
        ex.sync();
        if (tid == 0) {       // (I + S_uu Su) du + S_uy dnu = z_u ;  S_yu Su du + S_yy dnu = z_y : Gaussian elimination, partial pivoting
            double *M = sh.M, *x = sh.x;
            for (int a = 0; a < NF; ++a) {
                for (int c = 0; c < N; ++c) { double v = 0; for (int k = 0; k < N; ++k) v += sh.S[a * NF + k] * sh.Su[k * N + c]; M[a * NF + c] = v + (a == c ? 1.0 : 0.0); }
                M[a * NF + N] = sh.S[a * NF + N];
                x[a] = sh.z[a];
            }
            bool singular = false;
            for (int c = 0; c < NF; ++c) {
                int piv = c;
                for (int r = c + 1; r < NF; ++r) if (fabs(M[r * NF + c]) > fabs(M[piv * NF + c])) piv = r;
                if (M[piv * NF + c] == 0.0) { singular = true; break; }
                if (piv != c) { for (int k = 0; k < NF; ++k) { const double tmp = M[c * NF + k]; M[c * NF + k] = M[piv * NF + k]; M[piv * NF + k] = tmp; } const double tmp = x[c]; x[c] = x[piv]; x[piv] = tmp; }
                const double ipiv = rcp(M[c * NF + c]);
                for (int r = c + 1; r < NF; ++r) {
                    const double f = M[r * NF + c] * ipiv;
                    for (int k = c; k < NF; ++k) M[r * NF + k] -= f * M[c * NF + k];
                    x[r] -= f * x[c];
                }
            }
            if (singular) sh.status = -5;
            else for (int r = NF - 1; r >= 0; --r) { double a = x[r]; for (int k = r + 1; k < NF; ++k) a -= M[r * NF + k] * x[k]; x[r] = fdiv(a, M[r * NF + r]); }
            for (int k = 0; k < N; ++k) { double v = 0; for (int l = 0; l < N; ++l) v += sh.Su[k * N + l] * x[l]; sh.su[k] = v; }
            sh.su[N] = x[N]; sh.dnu = x[N];
        }
        ex.sync();
        ex.stamp(7);       // z, the small system (thread 0)
        for (int idx = tid; idx < L * NW; idx += nt) {
            const int e = idx / NW, k = idx - e * NW;
            const double *Xe = X + (size_t)e * NW * NF;
            double a = hr[idx];
            for (int c = 0; c < NF; ++c) a -= Xe[k * NF + c] * sh.su[c];
            dp[idx] = a;
        }
        ex.sync();
        for (int idx = tid; idx < L * R; idx += nt) {
            const int e = idx / R, r = idx - e * R, i = r / m, j = r - i * m;
            double a = dp[e * NW + 2 * N] * b[j];
            for (int k = 0; k < N; ++k) a -= A[j * N + k] * dp[e * NW + i * N + k];
            ds[idx] = a;
        }
        for (int k = tid; k < N; k += nt) { double a = 0; for (int e = 0; e < L; ++e) a += dp[e * NW + k] - dp[e * NW + N + k]; sh.dssoc[1 + k] = a; }
        ex.sync();
        if (tid == 0) {
            double a = -gt;
            for (int k = 0; k < N; ++k) a -= sh.cv[k] * sh.dssoc[1 + k];
            sh.dt = fdiv(a, sh.c0); sh.dssoc[0] = sh.dt;
        }
        ex.sync();
        ex.stamp(8);       // dp, ds, du, dt
    };
    // the rows of a Newton step: dl = kv - lam - (lam / s) ds; step bound, the two sums of the step-length model; prod: leave ds dl in kap
    auto rows = [&](bool corrector, double &amax, double &c1, double &c2) {
        double am = 1e300, s1 = 0, s2 = 0;
        for (int idx = tid; idx < L * R; idx += nt) {
            const double sv = sl[idx], lv = lam[idx], dsv = ds[idx], kv = corrector ? kap[idx] : 0.0;
            const double dlv = kv - lv - lv / sv * dsv;
            dl[idx] = dlv;
            if (dsv < 0) am = fmin(am, -sv / dsv);
            if (dlv < 0) am = fmin(am, -lv / dlv);
            s1 += sv * dlv + lv * dsv; s2 += dsv * dlv;
            if (!corrector) kap[idx] = dsv * dlv;
        }
        ex.reduce3(am, s1, s2);       // min, sum, sum over the workgroup in one pass
        amax = am; c1 = s1; c2 = s2;
    };
    for (it = 0; it <= P.ipm_max_iter; ++it) {
        const bool first_warm = use_warm && it == 0;       // the re-centring Newton step of a warm solve: no predictor, no stop test
        // ---- slacks, complementarity
        double gsum = 0;
        bool bad = false;
        for (int idx = tid; idx < L * R; idx += nt) {
            const int e = idx / R, r = idx - e * R, i = r / m, j = r - i * m;
            double a = pp[e * NW + 2 * N] * b[j];
            for (int k = 0; k < N; ++k) a -= A[j * N + k] * pp[e * NW + i * N + k];
            sl[idx] = a;
            if (!(a > 0)) bad = true;
            if (it == 0 && !use_warm) lam[idx] = 1.0 / a;
            gsum += a * lam[idx];
            dl[idx] = lam[idx] / a;         // the row's weight in the block Hessians (dl is free until the first Newton step)
        }
        for (int k = tid; k < N; k += nt) { double a = 0; for (int e = 0; e < L; ++e) a += pp[e * NW + k] - pp[e * NW + N + k]; sh.ssoc[1 + k] = a; }
        double flag = bad ? -1.0 : 1.0, unused = 0.0;
        ex.reduce3(flag, gsum, unused);
        const double gap_rows = gsum;
        const bool any_bad = flag < 0.0;
        ex.stamp(0);       // slacks + reduction
        if (tid == 0) {
            sh.ssoc[0] = sh.t;
            if (any_bad || !gcs_math::soc_interior<Q>(sh.ssoc)) { sh.status = -3; sh.stop = 1; }
            else {
                if (it == 0 && !use_warm) sh.lsoc[0] = 1.0 / sh.t;
                double gap = gap_rows;
                for (int k = 0; k < Q; ++k) gap += sh.ssoc[k] * sh.lsoc[k];
                sh.gap = gap; sh.mu = gap / deg;
                // the record the next solve restarts from: the first iterate, after at least one Newton step, with mu <= SAVE * mu_ref
                sh.save_now = rec != nullptr && !sh.saved && it >= 1 && sh.mu <= TERM_WS_SAVE * sh.mu_ref;
                if (sh.save_now) { sh.saved = 1; rec[0] = 1.0; rec[1] = P.rho; rec[2] = (double)L; rec[3] = sh.nu; }
                if (!first_warm && (sh.mu <= P.ipm_tol || (sh.stalled && sh.mu <= 1e3 * P.ipm_tol))) {
                    sh.status = (use_warm && !(sh.mu <= P.ipm_tol)) ? -7 : 0;      // (a warm solve does not leave through the precision-exhausted rule)
                    sh.stop = 1;
                }
                else if (it == P.ipm_max_iter) sh.stop = 1;
                else {
                    // Nesterov-Todd scaling of the cone pair, W^{-2}, the scaled point; t eliminated in closed form:
                    // W^{-2} = [c0 cv'; cv Mu],  Su = Mu - cv cv' / c0 = eta^-2 (I - 2 wb1 wb1' / (2 wb0^2 - 1))
                    const double ss = gcs_math::soc_det<Q>(sh.ssoc), zz = gcs_math::soc_det<Q>(sh.lsoc);
                    if (!(ss > 0.0) || !(zz > 0.0)) { sh.status = (sh.mu <= 1e3 * P.ipm_tol && !use_warm) ? 0 : -4; sh.stop = 1; }
                    else {
                        const double is = gcs_math::rsqrt_nr(ss), iz = gcs_math::rsqrt_nr(zz);
                        double dot = 0, wb[Q];
                        for (int k = 0; k < Q; ++k) dot += (sh.ssoc[k] * is) * (sh.lsoc[k] * iz);
                        const double gam = sqrt_nr(0.5 * (1.0 + dot));
                        const double i2g = 0.5 * rcp(gam);
                        wb[0] = (sh.ssoc[0] * is + sh.lsoc[0] * iz) * i2g;
                        for (int k = 1; k < Q; ++k) wb[k] = (sh.ssoc[k] * is - sh.lsoc[k] * iz) * i2g;
                        const double eta = sqrt_nr(sqrt_nr(fdiv(ss, zz))), ieta = rcp(eta), iw0 = rcp(1.0 + wb[0]);
                        for (int i = 0; i < Q; ++i)
                            for (int j = 0; j < Q; ++j) {
                                double w;
                                if (i == 0 && j == 0) w = wb[0];
                                else if (i == 0) w = wb[j];
                                else if (j == 0) w = wb[i];
                                else w = (i == j ? 1.0 : 0.0) + wb[i] * wb[j] * iw0;
                                sh.Wsoc[i * Q + j] = eta * w;
                                sh.Wsoci[i * Q + j] = (((i == 0) != (j == 0)) ? -w : w) * ieta;
                            }
                        for (int i = 0; i < Q; ++i)
                            for (int j = 0; j < Q; ++j) { double a = 0; for (int k = 0; k < Q; ++k) a += sh.Wsoci[i * Q + k] * sh.Wsoci[k * Q + j]; sh.W2[i * Q + j] = a; }
                        for (int i = 0; i < Q; ++i) { double a = 0; for (int k = 0; k < Q; ++k) a += sh.Wsoc[i * Q + k] * sh.lsoc[k]; sh.lt[i] = a; }
                        const double ie2 = ieta * ieta, g2 = 2.0 * rcp(2.0 * wb[0] * wb[0] - 1.0);
                        sh.c0 = ie2 * (2.0 * wb[0] * wb[0] - 1.0);
                        for (int k = 0; k < N; ++k) sh.cv[k] = -ie2 * 2.0 * wb[0] * wb[1 + k];
                        for (int k = 0; k < N; ++k) for (int l = 0; l < N; ++l) sh.Su[k * N + l] = ie2 * ((k == l ? 1.0 : 0.0) - g2 * wb[1 + k] * wb[1 + l]);
                    }
                }
            }
        }
        ex.sync();
        ex.stamp(1);       // stop test + cone scaling (thread 0)
        if (sh.save_now) {
            for (int idx = tid; idx < L * NW; idx += nt) { const int e = idx / NW, k = idx - e * NW; double *w = rec + TERM_W_HDR + e * W_PER; w[k] = pp[idx]; w[NW + k] = tg[idx]; }
            for (int idx = tid; idx < L * R; idx += nt) { const int e = idx / R, r = idx - e * R; rec[TERM_W_HDR + e * W_PER + 2 * NW + r] = lam[idx]; }
        }
        if (sh.stop) break;
        // ---- block Hessians  H_e = Q_e + REG + sum_rows (lam / s) g g',  g = (a_j on [O]_i, -b_j on y)
        for (int idx = tid; idx < L * NW * NW; idx += nt) {
            const int e = idx / (NW * NW), ac = idx - e * NW * NW, a = ac / NW, c = ac - a * NW;
            double v = a == c ? qd[e * NW + a] + TERM_REG : 0.0;
            const int ia = a < N ? 0 : (a < 2 * N ? 1 : 2), ic = c < N ? 0 : (c < 2 * N ? 1 : 2);
            const double *De = dl + e * R;
            if (ia == 2 && ic == 2) {
                for (int r = 0; r < R; ++r) { const int j = r < m ? r : r - m; v += De[r] * b[j] * b[j]; }
            } else if (ia == 2 || ic == 2) {
                const int i = ia == 2 ? ic : ia, k = (ia == 2 ? c : a) - i * N;
                for (int j = 0; j < m; ++j) v -= De[i * m + j] * A[j * N + k] * b[j];
            } else if (ia == ic) {
                const int ka = a - ia * N, kc = c - ia * N;
                for (int j = 0; j < m; ++j) v += De[ia * m + j] * A[j * N + ka] * A[j * N + kc];
            }
            H[idx] = v;
        }
        ex.sync();
        ex.stamp(2);       // block Hessians
        // ---- per block: Cholesky in place (clamped pivots; the diagonal holds 1 / L_jj: the solves multiply), then X = H^{-1} F', one
        //      (block, column) per thread
        for (int e = tid; e < L; e += nt) {
            double *Le = H + (size_t)e * NW * NW;
            for (int j = 0; j < NW; ++j) {
                const double d0 = Le[j * NW + j];       // (column j is untouched until now: the entry of H itself)
                double dj = d0;
                for (int k = 0; k < j; ++k) dj -= Le[j * NW + k] * Le[j * NW + k];
                if (!(dj > TERM_CHOL_SKIP * d0)) dj = d0 > 0 ? TERM_CHOL_SKIP * d0 : 1.0;
                const double inv = 1.0 / sqrt(dj);
                Le[j * NW + j] = inv;
                for (int i = j + 1; i < NW; ++i) {
                    double sv = Le[i * NW + j];
                    for (int k = 0; k < j; ++k) sv -= Le[i * NW + k] * Le[j * NW + k];
                    Le[i * NW + j] = sv * inv;
                }
            }
        }
        ex.sync();
        ex.stamp(3);       // Cholesky
        for (int idx = tid; idx < L * NF; idx += nt) {
            const int e = idx / NF, c = idx - e * NF;
            const double *Le = H + (size_t)e * NW * NW;
            double *Xe = X + (size_t)e * NW * NF;
            for (int k = 0; k < NW; ++k) Xe[k * NF + c] = 0.0;
            if (c < N) { Xe[c * NF + c] = 1.0; Xe[(N + c) * NF + c] = -1.0; } else Xe[2 * N * NF + c] = 1.0;
            for (int i = 0; i < NW; ++i) { double a = Xe[i * NF + c]; for (int k = 0; k < i; ++k) a -= Le[i * NW + k] * Xe[k * NF + c]; Xe[i * NF + c] = a * Le[i * NW + i]; }
            for (int i = NW - 1; i >= 0; --i) { double a = Xe[i * NF + c]; for (int k = i + 1; k < NW; ++k) a -= Le[k * NW + i] * Xe[k * NF + c]; Xe[i * NF + c] = a * Le[i * NW + i]; }
        }
        ex.sync();
        for (int ac = tid; ac < NF * NF; ac += nt) {      // S = sum_e F X_e
            const int a = ac / NF, c = ac - a * NF;
            double v = 0;
            for (int e = 0; e < L; ++e) {
                const double *Xe = X + (size_t)e * NW * NF;
                v += a < N ? Xe[a * NF + c] - Xe[(N + a) * NF + c] : Xe[2 * N * NF + c];
            }
            sh.S[ac] = v;
        }
        // ---- predictor
        for (int idx = tid; idx < L * R; idx += nt) kap[idx] = 0.0;
        if (tid == 0) for (int k = 0; k < Q; ++k) sh.ksoc[k] = 0.0;
        ex.sync();
        ex.stamp(4);       // X, S
        double amax, c1, c2;
        if (first_warm) {          // kappa = mu_ref / s, the cone's = mu_ref s^{-1}: no predictor, no second-order term
            if (tid == 0) {
                sh.sm = sh.mu_ref;
                const double idets = rcp(gcs_math::soc_det<Q>(sh.ssoc));
                for (int i = 0; i < Q; ++i) sh.ksoc[i] = sh.sm * (i == 0 ? sh.ssoc[0] : -sh.ssoc[i]) * idets;
            }
        } else {
        newton();
        rows(false, amax, c1, c2);
        ex.stamp(10);      // rows of the predictor
        if (tid == 0) {
            for (int i = 0; i < Q; ++i) { double a = -sh.lsoc[i]; for (int k = 0; k < Q; ++k) a -= sh.W2[i * Q + k] * sh.dssoc[k]; sh.dlsoc[i] = a; }
            amax = fmin(amax, fmin(gcs_math::soc_max_step<Q>(sh.ssoc, sh.dssoc), gcs_math::soc_max_step<Q>(sh.lsoc, sh.dlsoc)));
            for (int k = 0; k < Q; ++k) { c1 += sh.ssoc[k] * sh.dlsoc[k] + sh.lsoc[k] * sh.dssoc[k]; c2 += sh.dssoc[k] * sh.dlsoc[k]; }
            const double al_aff = fmin(1.0, amax);
            double sig = fdiv(sh.gap + al_aff * c1 + al_aff * al_aff * c2, deg * sh.mu);
            sig = sig < 0 ? 0 : (sig > 1 ? 1 : sig);
            sh.sm = sig * sig * sig * sh.mu;
            // cone: kappa = sigma mu s^{-1} - W^{-1} (lt \ ((W^{-1} ds_a) o (W dl_a)))
            double a1[Q], a2[Q], pr[Q], qv[Q];
            for (int i = 0; i < Q; ++i) {
                double u1 = 0, u2 = 0;
                for (int k = 0; k < Q; ++k) { u1 += sh.Wsoci[i * Q + k] * sh.dssoc[k]; u2 += sh.Wsoc[i * Q + k] * sh.dlsoc[k]; }
                a1[i] = u1; a2[i] = u2;
            }
            soc_prod<Q>(a1, a2, pr);
            soc_div<Q>(sh.lt, pr, qv);
            const double idets = rcp(gcs_math::soc_det<Q>(sh.ssoc));
            for (int i = 0; i < Q; ++i) {
                double a = 0;
                for (int k = 0; k < Q; ++k) a += sh.Wsoci[i * Q + k] * qv[k];
                sh.ksoc[i] = sh.sm * (i == 0 ? sh.ssoc[0] : -sh.ssoc[i]) * idets - a;
            }
        }
        }
        ex.sync();
        ex.stamp(11);      // sigma, cone multipliers (thread 0)
        // ---- corrector
        for (int idx = tid; idx < L * R; idx += nt) kap[idx] = (sh.sm - kap[idx]) / sl[idx];
        ex.sync();
        newton();
        rows(true, amax, c1, c2);
        ex.stamp(12);      // kappa + rows of the corrector
        if (tid == 0) {
            for (int i = 0; i < Q; ++i) { double a = sh.ksoc[i] - sh.lsoc[i]; for (int k = 0; k < Q; ++k) a -= sh.W2[i * Q + k] * sh.dssoc[k]; sh.dlsoc[i] = a; }
            amax = fmin(amax, fmin(gcs_math::soc_max_step<Q>(sh.ssoc, sh.dssoc), gcs_math::soc_max_step<Q>(sh.lsoc, sh.dlsoc)));
            double al = fmin(1.0, 0.99 * amax);
            for (int tries = 0; tries < 40; ++tries) {      // keep both cone points strictly inside despite round-off
                double s2[Q], l2[Q];
                for (int k = 0; k < Q; ++k) { s2[k] = sh.ssoc[k] + al * sh.dssoc[k]; l2[k] = sh.lsoc[k] + al * sh.dlsoc[k]; }
                if (gcs_math::soc_interior<Q>(s2) && gcs_math::soc_interior<Q>(l2)) break;
                al *= 0.7;
            }
            sh.al = al; sh.stalled = al < 1e-3;
            sh.t += al * sh.dt; sh.nu += al * sh.dnu;
            for (int k = 0; k < Q; ++k) sh.lsoc[k] += al * sh.dlsoc[k];
            if (sh.status == -5) sh.stop = 1;
        }
        ex.sync();
        if (sh.stop) break;
        const double al = sh.al;
        for (int idx = tid; idx < L * NW; idx += nt) pp[idx] += al * dp[idx];
        for (int idx = tid; idx < L * R; idx += nt) lam[idx] += al * dl[idx];
        ex.sync();
        ex.stamp(13);      // step length (thread 0) + update
    }
    it_total += it;
    ex.sync();
    const bool failed = sh.status != 0;
    ex.sync();
    if (tid == 0 && failed) {
        if (rec != nullptr) rec[0] = 0.0;      // no restart from a solve that failed
        sh.use_warm = 0;
    }
    ex.sync();
    if (!failed || !use_warm) break;
  }
    const int status = sh.status;
    if (status != 0) return status < -1 ? status : -1;      // (the copies of a failed solve keep their previous values)
    // ---- outputs: copies of every incidence (the dead side: y = 0, O = 0; the free word of an incoming edge sits at its target), x = z = sum O, y_v = 1
    for (int idx = tid; idx < P.d * NW; idx += nt) {
        const int e = idx / NW, k = idx - e * NW;
        const bool live = e >= lo && e < hi, outgoing = e >= P.d_in;
        const double *pe = pp + (e - lo) * NW;
        double v;
        if (k < N) v = outgoing ? (live ? pe[k] : 0.0) : target<N, T>(P, k, e);
        else if (k < 2 * N) v = live ? (outgoing ? pe[k] : pe[k - N]) : 0.0;
        else v = live ? pe[2 * N] : 0.0;
        P.copy[(size_t)k * P.NI + state_column<N, T>(P, e)] = (T)v;
    }
    for (int k = tid; k < 2 * N; k += nt) {
        double a = 0;
        for (int e = 0; e < L; ++e) a += pp[e * NW + k];
        P.xv[k] = a; P.zv[k] = a;
    }
    if (tid == 0) P.yv[0] = 1.0;
    return it_total;
}

}  // namespace gcs_term
