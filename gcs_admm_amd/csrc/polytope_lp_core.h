// polytope_lp_core.h -- the per-lane LP solver of polytope_lp.hip (host + device: the host build exists for
// the lock-step debugging harness under tests/hostemu only; the product path is the HIP kernels).
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define GCS_LP_HD __host__ __device__ __forceinline__
#else
#define GCS_LP_HD inline
#endif

namespace gcsadmm_lp {

constexpr int WAVE = 64;
constexpr int MAX_IT = 80;
constexpr double R_CAP = 1e6;      // the inscribed radius is capped (unbounded sets)
constexpr double X_CAP = 1e8;      // coordinates are capped in the bounding LPs (unbounded directions)

GCS_LP_HD constexpr int PK(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

struct Polys {
    int n, P;
    const int *ptr;        // [P+1]
    const double *A;       // [rows][n]
    const double *b;       // [rows]
    const double *nrm;     // [rows]  |a_i|_2
};

// one LP:  min c'w  s.t.  g_i'w <= h_i ;  rows = rows of polytope p (+ polytope q) (+ caps)
template <int N, bool BALL> struct Rows {
    static constexpr int K = BALL ? N + 1 : N;
    const Polys &S;
    int p0, m1, q0, m2, m;
    GCS_LP_HD Rows(const Polys &s, int p, int q) : S(s)
    {
        p0 = s.ptr[p]; m1 = s.ptr[p + 1] - p0;
        q0 = q >= 0 ? s.ptr[q] : 0; m2 = q >= 0 ? s.ptr[q + 1] - q0 : 0;
        m = m1 + m2 + (BALL ? 1 : 2 * N);
    }
    GCS_LP_HD void get(int i, double (&g)[K], double &h) const
    {
        if (i < m1 + m2) {
            const int r = i < m1 ? p0 + i : q0 + (i - m1);
#pragma unroll
            for (int k = 0; k < N; ++k) g[k] = S.A[(size_t)r * N + k];
            if (BALL) g[K - 1] = S.nrm[r];
            h = S.b[r];
        } else if (BALL) {
#pragma unroll
            for (int k = 0; k < N; ++k) g[k] = 0.0;
            g[K - 1] = 1.0; h = R_CAP;
        } else {
            const int j = i - (m1 + m2), k = j >> 1;
#pragma unroll
            for (int kk = 0; kk < N; ++kk) g[kk] = 0.0;
            g[k] = (j & 1) ? -1.0 : 1.0; h = X_CAP;
        }
    }
};

template <int K> GCS_LP_HD bool chol(double (&H)[K * (K + 1) / 2])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        double d = H[PK(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= H[PK(j, k)] * H[PK(j, k)];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double inv = 1.0 / sqrt(d);
        H[PK(j, j)] = inv;
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            double s = H[PK(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= H[PK(i, k)] * H[PK(j, k)];
            H[PK(i, j)] = s * inv;
        }
    }
    return ok;
}
template <int K> GCS_LP_HD void chol_solve(const double (&L)[K * (K + 1) / 2], double (&x)[K])
{
#pragma unroll
    for (int i = 0; i < K; ++i) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[PK(i, k)] * x[k];
        x[i] = s * L[PK(i, i)];
    }
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < K; ++k) s -= L[PK(k, i)] * x[k];
        x[i] = s * L[PK(i, i)];
    }
}

// status: 0 converged, 1 stopped early with r > 0, 2 stopped early with the dual bound below -tol, -1 iteration limit
template <int N, bool BALL>
GCS_LP_HD int lp_ipm(const Rows<N, BALL> &R, const double (&c)[Rows<N, BALL>::K], double (&w)[Rows<N, BALL>::K],
                      double *lam, double *dlam, int lane, bool early, double tol, int *iters_out)
{
    constexpr int K = Rows<N, BALL>::K, KS = K * (K + 1) / 2;
    const int m = R.m;
    double g[K], h;
    for (int i = 0; i < m; ++i) {            // duals on the central path of the start: lam = 1 / s
        R.get(i, g, h);
        double s = h;
#pragma unroll
        for (int k = 0; k < K; ++k) s -= g[k] * w[k];
        lam[i * WAVE + lane] = 1.0 / s;
    }
    int status = -1, it = 0;
    for (; it < MAX_IT; ++it) {
        double H[KS], rd[K], gap = 0, hl = 0;
#pragma unroll
        for (int k = 0; k < KS; ++k) H[k] = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) rd[k] = c[k];
        for (int i = 0; i < m; ++i) {
            R.get(i, g, h);
            double s = h;
#pragma unroll
            for (int k = 0; k < K; ++k) s -= g[k] * w[k];
            const double l = lam[i * WAVE + lane], d = l / s;
            gap += s * l; hl += h * l;
#pragma unroll
            for (int a = 0; a < K; ++a) {
                rd[a] += l * g[a];
#pragma unroll
                for (int b2 = 0; b2 <= a; ++b2) H[PK(a, b2)] += d * g[a] * g[b2];
            }
        }
        const double mu = gap / m;
        double rdmax = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) rdmax = fmax(rdmax, fabs(rd[k]));
        if (BALL && early) {
            if (w[K - 1] > 0.0) { status = 1; break; }                       // a point with a ball around it: they overlap
            // dual bound r* <= h'lam, valid up to the dual residual times |w*|: only used with a clear margin,
            // near-touching pairs run to convergence and are decided on r* itself
            if (rdmax <= 1e-9 && hl < -tol - 1e-6) { status = 2; break; }
        }
        if (mu <= 1e-11 * fmax(1.0, fabs(BALL ? w[K - 1] : 1.0)) && rdmax <= 1e-9) { status = 0; break; }
        double tr = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) tr += H[PK(k, k)];
#pragma unroll
        for (int k = 0; k < K; ++k) H[PK(k, k)] += 1e-15 * tr;
        chol<K>(H);
        double dwa[K];
#pragma unroll
        for (int k = 0; k < K; ++k) dwa[k] = -c[k];
        chol_solve<K>(H, dwa);
        // affine step: bound and the complementarity it would leave
        double amax = 1e300, c1 = 0, c2 = 0;
        for (int i = 0; i < m; ++i) {
            R.get(i, g, h);
            double s = h, q = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) { s -= g[k] * w[k]; q += g[k] * dwa[k]; }
            const double l = lam[i * WAVE + lane], d = l / s;
            const double dsa = -q, dla = -l + d * q;
            if (dsa < 0) amax = fmin(amax, -s / dsa);
            if (dla < 0) amax = fmin(amax, -l / dla);
            c1 += s * dla + l * dsa; c2 += dsa * dla;
        }
        const double ala = fmin(1.0, amax);
        double sig = (gap + ala * c1 + ala * ala * c2) / gap;
        sig = sig < 0 ? 0 : (sig > 1 ? 1 : sig);
        sig = sig * sig * sig;
        const double sm = sig * mu;
        double dw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) dw[k] = -c[k];
        for (int i = 0; i < m; ++i) {
            R.get(i, g, h);
            double s = h, q = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) { s -= g[k] * w[k]; q += g[k] * dwa[k]; }
            const double l = lam[i * WAVE + lane], d = l / s;
            const double dsa = -q, dla = -l + d * q;
            const double f = (sm - dsa * dla) / s;
#pragma unroll
            for (int k = 0; k < K; ++k) dw[k] -= g[k] * f;
        }
        chol_solve<K>(H, dw);
        amax = 1e300;
        for (int i = 0; i < m; ++i) {
            R.get(i, g, h);
            double s = h, q = 0, qa = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) { s -= g[k] * w[k]; q += g[k] * dw[k]; qa += g[k] * dwa[k]; }
            const double l = lam[i * WAVE + lane], d = l / s;
            const double dsa = -qa, dla = -l + d * qa;
            const double ds = -q, dl = (sm - dsa * dla) / s - l + d * q;
            if (ds < 0) amax = fmin(amax, -s / ds);
            if (dl < 0) amax = fmin(amax, -l / dl);
            dlam[i * WAVE + lane] = dl;
        }
        const double al = fmin(1.0, 0.99 * amax);
        if (!(al > 1e-10)) {             // no progress (or a slack lost to round-off): the point is as good as f64 gets
            if (mu <= 1e-7 * fmax(1.0, fabs(BALL ? w[K - 1] : 1.0)) && rdmax <= 1e-7) status = 0;
            break;
        }
        for (int i = 0; i < m; ++i) lam[i * WAVE + lane] += al * dlam[i * WAVE + lane];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] += al * dw[k];
    }
    if (iters_out) *iters_out = it;
    return status;
}

// strictly feasible start of the ball LP from a point x0: r0 one unit below the tightest row
template <int N> GCS_LP_HD void ball_start(const Rows<N, true> &R, const double *x0, double (&w)[N + 1])
{
    double g[N + 1], h, r0 = R_CAP - 1.0;
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = x0 ? x0[k] : 0.0;
    if (!x0) {   // no start point given: the least-squares point of A x = b (near the middle of a bounded region,
                 // wherever it sits; starting at the origin costs iterations and digits when the region is far away)
        double G[N * (N + 1) / 2], t[N];
#pragma unroll
        for (int k = 0; k < N * (N + 1) / 2; ++k) G[k] = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = 0;
        for (int i = 0; i < R.m - 1; ++i) {
            R.get(i, g, h);
#pragma unroll
            for (int a = 0; a < N; ++a) {
                t[a] += g[a] * h;
#pragma unroll
                for (int b2 = 0; b2 <= a; ++b2) G[PK(a, b2)] += g[a] * g[b2];
            }
        }
        double tr = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) tr += G[PK(k, k)];
#pragma unroll
        for (int k = 0; k < N; ++k) G[PK(k, k)] += 1e-12 * tr;
        if (chol<N>(G)) {
            chol_solve<N>(G, t);
#pragma unroll
            for (int k = 0; k < N; ++k) w[k] = t[k];
        }
    }
    for (int i = 0; i < R.m - 1; ++i) {
        R.get(i, g, h);
        double s = h;
#pragma unroll
        for (int k = 0; k < N; ++k) s -= g[k] * w[k];
        r0 = fmin(r0, s / g[N]);
    }
    w[N] = r0 - 1.0;
}


} // namespace gcsadmm_lp
