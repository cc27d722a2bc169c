// vertex_program.h -- the per-wavefront program of the x-update (vertex step) kernel.
//
// One 64-lane wavefront (= one workgroup) solves several vertex sub-problems at once.
// A vertex with d incident edges occupies d+1 consecutive lanes ("group"):
//     lane 0 of the group  = the BORDER lane: unknowns (z_v, y_v) with facet rows 1,2
//                            (reference admm_solver_v3.py:420-426), the cone t >= |z_v1 - z_v2|
//                            (:380-384), x_v, and all group-level linear algebra;
//     lanes 1..d           = one BLOCK lane per incident edge: unknowns (O_e, y_e) with facet
//                            rows 3,4 (:434-440) and the consensus penalty (:392-413).
// Rows 1,2 have exactly the shape of rows 3,4 with (z_v, y_v) in place of (O_e, y_e), so border
// and block lanes run the same row passes in lock step.  Constraints 5-7 (:443-464) are the
// reductions of SURVEY.md Appendix A.3 plus the two flow equalities, eliminated through the
// per-side sums B_in / B_out.
//
// The program is written as barrier-separated PHASES over a per-lane state struct, so that the
// same code runs as a HIP kernel (state in registers, LDS = __shared__) and, for debugging on a
// machine without a GPU, as a lock-step host emulation (tests/hostemu, never shipped or timed).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GCS_HD __host__ __device__ __forceinline__
#else
#define GCS_HD inline
#endif

namespace gcs {

constexpr int WAVE = 64;
constexpr int MAX_SLOTS = 8;     // vertices per wavefront
constexpr int RED_CHUNK = 28;    // rows of the reduction staging area (>= the largest round, checked below)
constexpr double CHOL_SKIP = 1e-12;
// Tikhonov term (REG_DELTA/2)|w|^2 on every centred unknown (see oracle/gcs_oracle.c REG_DELTA)
constexpr double REG_DELTA = 1e-7;

// reciprocal: on the device the hardware estimate refined by two Newton steps (the IEEE division
// sequence costs ~3x as many instructions and the kernel does ~100 of these per Newton iteration)
GCS_HD double rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

GCS_HD constexpr int PK(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

template <int N> struct Dim {
    static constexpr int N2 = 2 * N, NW = 2 * N + 1, NB = 4 * N + 2, Q = N + 1;
    static constexpr int NB1 = NB - 1;   // border unknowns without the epigraph variable t (eliminated by hand)
    static constexpr int NS = N * (N + 1) / 2, NWS = NW * (NW + 1) / 2, NBS = NB1 * (NB1 + 1) / 2,
                         N2S = N2 * (N2 + 1) / 2;
    // reduction round 1 layout
    // three chunks, each computed right before it is staged:
    //   A: B_e, the block's unknowns, complementarity, X_ii   B: B_e X_e, B_e(-g)   C: X'BX, X'B(-g)
    static constexpr int R1_B = 0, R1_W = NWS, R1_GAP = R1_W + NW, R1_XII = R1_GAP + 1, R1A_N = R1_XII + 2 * NS,
                         R1_BX = R1A_N, R1_T = R1_BX + NW * N2, R1B_N = NW * N2 + NW,
                         R1_XBX = R1_T + NW, R1_XT = R1_XBX + N2S, R1C_N = N2S + N2, R1_N = R1_XT + N2;
    static constexpr int R2_BASE = R1_N, R2_N = 3;                 // amax (min), c1, c2
    static constexpr int R3_BASE = R2_BASE + R2_N;                 // corrector right-hand sides
    static constexpr int R3_T = 0, R3_XT = NW, R3_GX = NW + N2, R3_N = NW + 2 * N2;
    static constexpr int R4_BASE = R3_BASE + R3_N, R4_N = 1;       // amax (min)
    static constexpr int R0_BASE = R4_BASE + R4_N, R0_N = 1;       // scale (max), once before the loop
    static constexpr int NSUM = R0_BASE + R0_N;
};

// per-slot (vertex) LDS layout, in doubles
template <int N> struct SlotLayout {
    using D = Dim<N>;
    int MM;
    int A, B, CEN, X, NU, DX, DXA, DNU, SC, SIN, SOUT, BSI, BXS, YS, MF, BORD, SIZE;
    GCS_HD explicit SlotLayout(int mm) : MM(mm)
    {
        int o = 0;
        A = o; o += mm * N;
        B = o; o += mm;
        CEN = o; o += N;
        X = o; o += 2 * D::N2;      // x_v (centred), ping-pong by iteration parity
        NU = o; o += 2 * D::NW;     // multipliers of the two flow equalities
        DX = o; o += D::N2;         // x part of the final Newton direction
        DXA = o; o += D::N2;        // x part of the affine direction
        DNU = o; o += 2 * D::NW;
        SC = o; o += 8;             // scalars: 0 alpha, 1 sigma*mu, 2 done, 3 mu, 4 scale, 5 status
        SIN = o; o += D::NSUM;      // reduced sums over the incoming / outgoing block lanes
        SOUT = o; o += D::NSUM;
        BSI = o; o += 2 * D::NWS;   // inverse of B_in / B_out
        BXS = o; o += 2 * D::NW * D::N2;
        YS = o; o += 2 * D::NW * D::N2;   // Bs^{-1} BXs per side
        MF = o; o += D::NBS;        // reduced border matrix, then its Cholesky factor
        BORD = o; o += 7 * D::Q + 3 * D::Q * D::Q + D::N2 + 2 * D::NB + N;   // border-lane state
        SIZE = (o + 1) & ~1;
    }
};

GCS_HD int lds_doubles(int n, int mm, int slots)
{
    // per-lane duals of rows a/b ( 2 halves x mm ) + 2 bounds, reduction staging, slots
    int per_lane = (4 * mm + 2) * WAVE;
    int stage = RED_CHUNK * WAVE;
    int slot = n == 2 ? SlotLayout<2>(mm).SIZE : SlotLayout<3>(mm).SIZE;
    return per_lane + stage + slots * slot;
}

// ---------------------------------------------------------------------------------------------
// small fixed-size dense helpers on packed lower-triangular storage (all indices compile-time)
// ---------------------------------------------------------------------------------------------
template <int NN> GCS_HD void chol_packed(double (&A)[NN * (NN + 1) / 2])
{
    // Pivots that have cancelled below CHOL_SKIP of their own diagonal entry are round-off, not
    // curvature: they are clamped to that floor (same rule as oracle/gcs_oracle.c chol()).
    double diag[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) diag[j] = A[PK(j, j)];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        double d = A[PK(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[PK(j, k)] * A[PK(j, k)];
        if (!(d > CHOL_SKIP * diag[j])) {
#if !defined(__HIPCC__) && defined(GCS_EMU_TRACE)
            if (getenv("GCS_EMU_TRACE")) fprintf(stderr, "clamp NN=%d j=%d d=%.3e diag=%.3e\n", NN, j, d, diag[j]);
#endif
            d = diag[j] > 0.0 ? CHOL_SKIP * diag[j] : 1.0;
        }
        const double inv = rcp(sqrt(d));
        A[PK(j, j)] = inv;          // the diagonal holds 1 / L_jj
#pragma unroll
        for (int i = j + 1; i < NN; ++i) {
            double s = A[PK(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= A[PK(i, k)] * A[PK(j, k)];
            A[PK(i, j)] = s * inv;
        }
    }
}
template <int NN> GCS_HD void chol_solve_packed(const double (&L)[NN * (NN + 1) / 2], double (&b)[NN])
{
#pragma unroll
    for (int i = 0; i < NN; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[PK(i, k)] * b[k];
        b[i] = s * L[PK(i, i)];
    }
#pragma unroll
    for (int i = NN - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < NN; ++k) s -= L[PK(k, i)] * b[k];
        b[i] = s * L[PK(i, i)];
    }
}
// the same factorisation / solve on a matrix that lives in LDS (only one row is cached in registers)
template <int NN> GCS_HD void chol_lds(double *A)
{
    double diag[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) diag[j] = A[PK(j, j)];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        double rowj[NN];
        double d = diag[j];
#pragma unroll
        for (int k = 0; k < j; ++k) { rowj[k] = A[PK(j, k)]; d -= rowj[k] * rowj[k]; }
        if (!(d > CHOL_SKIP * diag[j])) d = diag[j] > 0.0 ? CHOL_SKIP * diag[j] : 1.0;
        const double inv = rcp(sqrt(d));
        A[PK(j, j)] = inv;
#pragma unroll
        for (int i = j + 1; i < NN; ++i) {
            double s = A[PK(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= A[PK(i, k)] * rowj[k];
            A[PK(i, j)] = s * inv;
        }
    }
}
template <int NN> GCS_HD void chol_solve_lds(const double *L, double (&b)[NN])
{
#pragma unroll
    for (int i = 0; i < NN; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[PK(i, k)] * b[k];
        b[i] = s * L[PK(i, i)];
    }
#pragma unroll
    for (int i = NN - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < NN; ++k) s -= L[PK(k, i)] * b[k];
        b[i] = s * L[PK(i, i)];
    }
}
template <int NN> GCS_HD void chol_inverse_packed(const double (&L)[NN * (NN + 1) / 2], double (&X)[NN * (NN + 1) / 2])
{
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        double e[NN];
#pragma unroll
        for (int i = 0; i < NN; ++i) e[i] = (i == j) ? 1.0 : 0.0;
        chol_solve_packed<NN>(L, e);
#pragma unroll
        for (int i = j; i < NN; ++i) X[PK(i, j)] = e[i];
    }
}

// ---------------------------------------------------------------------------------------------
// second-order cone of dimension Q = N+1
// ---------------------------------------------------------------------------------------------
template <int Q> GCS_HD double soc_det(const double *s)
{
    double nn = 0;
#pragma unroll
    for (int k = 1; k < Q; ++k) nn += s[k] * s[k];
    nn = sqrt(nn);
    return (s[0] - nn) * (s[0] + nn);
}
template <int Q> GCS_HD bool soc_interior(const double *s)
{
    double nn = 0;
#pragma unroll
    for (int k = 1; k < Q; ++k) nn += s[k] * s[k];
    return s[0] > sqrt(nn);
}
template <int Q> GCS_HD double soc_max_step(const double *s, const double *ds)
{
    double a = ds[0] * ds[0], b = s[0] * ds[0];
    const double c = soc_det<Q>(s);
#pragma unroll
    for (int k = 1; k < Q; ++k) { a -= ds[k] * ds[k]; b -= s[k] * ds[k]; }
    b *= 2;
    double al = 1e300;
    if (ds[0] < 0) al = fmin(al, -s[0] / ds[0]);
    if (fabs(a) < 1e-300) {
        if (b < 0) al = fmin(al, -c / b);
    } else {
        const double disc = b * b - 4 * a * c;
        if (disc >= 0) {
            const double sq = sqrt(disc);
            const double qq = -0.5 * (b + (b >= 0 ? sq : -sq));
            const double r1 = qq / a, r2 = (qq != 0.0) ? c / qq : 1e300;
            if (r1 > 0) al = fmin(al, r1);
            if (r2 > 0) al = fmin(al, r2);
        }
    }
    return al;
}
// Nesterov-Todd scaling W (symmetric, full QxQ): W lam = W^{-1} s.  Returns false on a boundary point.
template <int Q> GCS_HD bool soc_scaling(const double *s, const double *z, double *W, double *Wi, double *wb, double &eta)
{
    const double ss = soc_det<Q>(s), zz = soc_det<Q>(z);
    if (!(ss > 0.0) || !(zz > 0.0)) return false;
    const double is = 1.0 / sqrt(ss), iz = 1.0 / sqrt(zz);
    double dot = 0;
#pragma unroll
    for (int k = 0; k < Q; ++k) dot += (s[k] * is) * (z[k] * iz);
    const double gam = sqrt(0.5 * (1.0 + dot));
    wb[0] = (s[0] * is + z[0] * iz) / (2 * gam);
#pragma unroll
    for (int k = 1; k < Q; ++k) wb[k] = (s[k] * is - z[k] * iz) / (2 * gam);
    eta = sqrt(sqrt(ss / zz));
#pragma unroll
    for (int i = 0; i < Q; ++i)
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            double w;
            if (i == 0 && j == 0) w = wb[0];
            else if (i == 0) w = wb[j];
            else if (j == 0) w = wb[i];
            else w = (i == j ? 1.0 : 0.0) + wb[i] * wb[j] / (1.0 + wb[0]);
            W[i * Q + j] = eta * w;
            Wi[i * Q + j] = (((i == 0) != (j == 0)) ? -w : w) / eta;
        }
    return true;
}

// ---------------------------------------------------------------------------------------------
// per-lane state
// ---------------------------------------------------------------------------------------------
enum Role : int { IDLE = 0, BORDER = 1, BLOCK = 2 };

template <int N> struct Lane {
    using D = Dim<N>;
    // identity
    int role, slot, v, d, d_in, glane, gbase, m;
    int inc, edge, out;            // block lanes: incidence slot, edge id, 1 = outgoing
    // unknowns of this lane: p = O_e (block) or z_v (border); yy = y_e or y_v
    double p[D::N2], yy;
    double dpa[D::NW], dp[D::NW];  // affine and final Newton directions of (p, yy)
    double T1[N], T2[N], Ty;       // block: targets (un-centred)
    double Tfree[N];               // incoming block: the coupled word that is only penalised
    // Hessian pieces of the rows owned by this lane (see pass_A)
    double K1[D::NS], K2[D::NS], k1y[N], k2y[N], kyy;
    double X1[D::NS], X2[D::NS], xy1[N], xy2[N];
    double g0[D::NW];              // gradient of the smooth objective part + equality multipliers
    double Bm[D::NWS];             // block: inverse of the block Hessian
    double P1[2 * D::N2 + 1], P2[2 * D::N2 + 1]; // G'(1/s) and G'(corr/s): parts p, y, x
    double gk[2 * D::N2 + 1];      // G' kappa
    double l5, l6;                 // duals of 0 <= y <= 1
    double amax, c1, c2, gap;
    // border lane only
    // (arrays live in the vertex's LDS slot: only one lane per vertex touches them, and keeping them
    //  in registers would charge every lane of the wavefront for them)
    double t, soc_c0, mu, scale;   // W^{-2} = [c0 cv'; cv C11]; t is eliminated by hand
    double *lsoc, *x, *W, *Wi, *W2, *lt, *soc_cv, *dba, *db, *dssa, *dlsa, *dss, *dls, *ksoc;
    int iters, status, done, stalled, bad;
    double BX[D::NW * D::N2], t0[D::NW];   // block: B_e X_e and B_e(-g0), kept from chunk B to chunk C
    double scale0;
};

struct WaveShared {
    double *lamA, *lamB;   // [2*MM][64] duals of rows a (1 or 3) and b (2 or 4)
    double *stage;         // [RED_CHUNK][64]
    double *slots;         // [MAX_SLOTS][slot size]
    int MM;
};

template <int N> GCS_HD double *slot_ptr(const WaveShared &S, const SlotLayout<N> &SL, int slot)
{
    return S.slots + (size_t)slot * SL.SIZE;
}

// ---------------------------------------------------------------------------------------------
// rows of one half i (compile-time) for a lane: F is called with (j, a[N], b, sa, sb, D? ...)
// sa = b*yy - a.p_i            (rows 1 / 3)
// sb = b*(1-yy) - a.(x_i - p_i) (rows 2 / 4)
// ---------------------------------------------------------------------------------------------
template <int N, int I, class F>
GCS_HD void rows_half(const Lane<N> &L, const double *A, const double *bc, const double *x, F &&f)
{
    double xi[N];   // loop invariant: read once (the stores of the passes could alias it for the compiler)
#pragma unroll
    for (int k = 0; k < N; ++k) xi[k] = x[I * N + k];
#pragma unroll 2
    for (int j = 0; j < L.m; ++j) {
        double a[N];
#pragma unroll
        for (int k = 0; k < N; ++k) a[k] = A[j * N + k];
        const double b = bc[j];
        double ap = 0, ax = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) { ap += a[k] * L.p[I * N + k]; ax += a[k] * xi[k]; }
        const double sa = b * L.yy - ap;
        const double sb = b * (1.0 - L.yy) - (ax - ap);
        f(j, a, b, sa, sb);
    }
}

// direction slacks of one row: dsa = b*dy - a.dp_i ; dsb = -b*dy - a.(dx_i - dp_i)
template <int N, int I>
GCS_HD void row_dir(const double (&a)[N], double b, const double (&dp)[2 * N + 1], const double *dx, double &dsa, double &dsb)
{
    double adp = 0, adx = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) { adp += a[k] * dp[I * N + k]; adx += a[k] * dx[I * N + k]; }
    dsa = b * dp[2 * N] - adp;
    dsb = -b * dp[2 * N] - (adx - adp);
}

// ---------------------------------------------------------------------------------------------
// PASS A: slacks, scalings, Hessian pieces, complementarity, G'lambda
// ---------------------------------------------------------------------------------------------
template <int N, int I>
GCS_HD void pass_A_half(Lane<N> &L, const double *A, const double *bc, const double *x, double *la, double *lb,
                        int MM, int lane, bool first, double (&Ki)[Dim<N>::NS], double (&kiy)[N],
                        double (&Xi)[Dim<N>::NS], double (&xyi)[N])
{
    rows_half<N, I>(L, A, bc, x, [&](int j, const double(&a)[N], double b, double sa, double sb) {
        double &ra = la[(I * MM + j) * WAVE + lane];
        double &rb = lb[(I * MM + j) * WAVE + lane];
        const double isa = rcp(sa), isb = rcp(sb);
        if (first) { ra = isa; rb = isb; }
        if (!(sa > 0.0) || !(sb > 0.0)) L.bad = 1;
        const double l_a = ra, l_b = rb;
        const double Da = l_a * isa, Db = l_b * isb, Ds = Da + Db;
#pragma unroll
        for (int k = 0; k < N; ++k) {
#pragma unroll
            for (int l = 0; l <= k; ++l) {
                Ki[PK(k, l)] += Ds * a[k] * a[l];
                Xi[PK(k, l)] += Db * a[k] * a[l];
            }
            kiy[k] -= Ds * b * a[k];
            xyi[k] += Db * b * a[k];
        }
        L.kyy += Ds * b * b;
        L.gap += sa * l_a + sb * l_b;
    });
}

template <int N>
GCS_HD void pass_A(Lane<N> &L, const WaveShared &S, const SlotLayout<N> &SL, int lane, bool first, int par, double rho, double eps_edge)
{
    using D = Dim<N>;
    if (L.role == IDLE || L.done) return;
    double *sl = slot_ptr<N>(S, SL, L.slot);
    const double *A = sl + SL.A, *bc = sl + SL.B, *cen = sl + SL.CEN, *x = sl + SL.X + par * D::N2;
#pragma unroll
    for (int k = 0; k < D::NS; ++k) { L.K1[k] = L.K2[k] = L.X1[k] = L.X2[k] = 0; }
#pragma unroll
    for (int k = 0; k < N; ++k) { L.k1y[k] = L.k2y[k] = L.xy1[k] = L.xy2[k] = 0; }
    L.kyy = 0; L.gap = 0; L.bad = 0;
    pass_A_half<N, 0>(L, A, bc, x, S.lamA, S.lamB, S.MM, lane, first, L.K1, L.k1y, L.X1, L.xy1);
    pass_A_half<N, 1>(L, A, bc, x, S.lamA, S.lamB, S.MM, lane, first, L.K2, L.k2y, L.X2, L.xy2);
    // bounds 0 <= yy <= 1
    {
        const double s5 = L.yy, s6 = 1.0 - L.yy;
        const double is5 = rcp(s5), is6 = rcp(s6);
        if (first) { L.l5 = is5; L.l6 = is6; }
        L.kyy += L.l5 * is5 + L.l6 * is6;
        L.gap += s5 * L.l5 + s6 * L.l6;
        if (!(s5 > 0.0) || !(s6 > 0.0)) L.bad = 1;
    }
    if (L.bad) L.gap = 0.0 / 0.0;   // poisons mu: the border lane stops this vertex with status -3
#pragma unroll
    for (int k = 0; k < N; ++k) { L.K1[PK(k, k)] += REG_DELTA; L.K2[PK(k, k)] += REG_DELTA; }
    L.kyy += REG_DELTA;
    if (L.role == BLOCK) {
        const double *nu = sl + SL.NU + (L.out ? D::NW : 0);
        double gy = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double g1 = rho * (L.p[k] + L.yy * cen[k] - L.T1[k]);
            const double g2 = L.out ? rho * (L.p[N + k] + L.yy * cen[k] - L.T2[k]) : 0.0;
            L.g0[k] = g1 - nu[k] + REG_DELTA * L.p[k];
            L.g0[N + k] = g2 - nu[N + k] + REG_DELTA * L.p[N + k];
            gy += cen[k] * (g1 + g2);
            L.K1[PK(k, k)] += rho;
            L.k1y[k] += rho * cen[k];
            L.kyy += rho * cen[k] * cen[k];
            if (L.out) {
                L.K2[PK(k, k)] += rho;
                L.k2y[k] += rho * cen[k];
                L.kyy += rho * cen[k] * cen[k];
            }
        }
        L.g0[2 * N] = rho * (L.yy - L.Ty) + eps_edge + gy - nu[2 * N] + REG_DELTA * L.yy;
        L.kyy += rho;
    } else {
        const double *nu = sl + SL.NU;
#pragma unroll
        for (int k = 0; k < D::NW; ++k) L.g0[k] = nu[k] + nu[D::NW + k] + REG_DELTA * (k < D::N2 ? L.p[k] : L.yy);
    }
}

// block lanes: factor the arrow Hessian [K1 0 k1y; 0 K2 k2y; . . kyy], explicit inverse Bm,
// products with X = d(omega)/d(x) coupling, and stage round-1 values
template <int N> struct BlockProducts {
    using D = Dim<N>;
    double BX[D::NW * D::N2];
    double XBX[D::N2S];
};

template <int N> GCS_HD void block_factor(Lane<N> &L)
{
    using D = Dim<N>;
    // Cholesky of the 2n+1 arrow matrix in the variable order (O_1, O_2, y): identical pivots to
    // the dense factorisation of the oracle because O_1 and O_2 are not directly coupled.
    double F1[D::NS], F2[D::NS];
#pragma unroll
    for (int k = 0; k < D::NS; ++k) { F1[k] = L.K1[k]; F2[k] = L.K2[k]; }
    chol_packed<N>(F1);
    chol_packed<N>(F2);
    // l_i = L_i^{-1} k_iy  (forward substitution only)
    double l1[N], l2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s1 = L.k1y[i], s2 = L.k2y[i];
#pragma unroll
        for (int k = 0; k < i; ++k) { s1 -= F1[PK(i, k)] * l1[k]; s2 -= F2[PK(i, k)] * l2[k]; }
        l1[i] = s1 * F1[PK(i, i)];
        l2[i] = s2 * F2[PK(i, i)];
    }
    double dy = L.kyy;
#pragma unroll
    for (int k = 0; k < N; ++k) dy -= l1[k] * l1[k] + l2[k] * l2[k];
    if (!(dy > CHOL_SKIP * L.kyy)) {
#if !defined(__HIPCC__) && defined(GCS_EMU_TRACE)
        if (getenv("GCS_EMU_TRACE")) fprintf(stderr, "clamp ypivot dy=%.3e kyy=%.3e\n", dy, L.kyy);
#endif
        dy = L.kyy > 0.0 ? CHOL_SKIP * L.kyy : 1.0;
    }
    const double isy = rcp(dy);
    // u_i = K_i^{-1} k_iy = L_i^{-T} l_i
    double u1[N], u2[N];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s1 = l1[i], s2 = l2[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) { s1 -= F1[PK(k, i)] * u1[k]; s2 -= F2[PK(k, i)] * u2[k]; }
        u1[i] = s1 * F1[PK(i, i)];
        u2[i] = s2 * F2[PK(i, i)];
    }
    double I1[D::NS], I2[D::NS];
    chol_inverse_packed<N>(F1, I1);
    chol_inverse_packed<N>(F2, I2);
    // B = [I1 + u1 u1'/sy, u1 u2'/sy, -u1/sy; . , I2 + u2 u2'/sy, -u2/sy; . . 1/sy]
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            L.Bm[PK(i, j)] = I1[PK(i, j)] + u1[i] * u1[j] * isy;
            L.Bm[PK(N + i, N + j)] = I2[PK(i, j)] + u2[i] * u2[j] * isy;
        }
#pragma unroll
        for (int j = 0; j < N; ++j) L.Bm[PK(N + i, j)] = u2[i] * u1[j] * isy;
        L.Bm[PK(2 * N, i)] = -u1[i] * isy;
        L.Bm[PK(2 * N, N + i)] = -u2[i] * isy;
    }
    L.Bm[PK(2 * N, 2 * N)] = isy;
}

// y = Bm * r
template <int N> GCS_HD void block_apply(const Lane<N> &L, const double (&r)[2 * N + 1], double (&y)[2 * N + 1])
{
    constexpr int NW = 2 * N + 1;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) s += L.Bm[PK(i, k)] * r[k];
        y[i] = s;
    }
}
// (X dx): X = [-X1 0; 0 -X2; xy1' xy2'] (rows O_1, O_2, y ; columns x_1, x_2)
template <int N> GCS_HD void X_apply(const Lane<N> &L, const double *dx, double (&y)[2 * N + 1])
{
    double sy = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int l = 0; l < N; ++l) { s1 -= L.X1[PK(k, l)] * dx[l]; s2 -= L.X2[PK(k, l)] * dx[N + l]; }
        y[k] = s1; y[N + k] = s2;
        sy += L.xy1[k] * dx[k] + L.xy2[k] * dx[N + k];
    }
    y[2 * N] = sy;
}
// X' t
template <int N> GCS_HD void XT_apply(const Lane<N> &L, const double (&t)[2 * N + 1], double (&y)[2 * N])
{
#pragma unroll
    for (int a = 0; a < N; ++a) {
        double s1 = L.xy1[a] * t[2 * N], s2 = L.xy2[a] * t[2 * N];
#pragma unroll
        for (int k = 0; k < N; ++k) { s1 -= L.X1[PK(a, k)] * t[k]; s2 -= L.X2[PK(a, k)] * t[N + k]; }
        y[a] = s1; y[N + a] = s2;
    }
}

// reduction round 1 of a block lane, written straight into the staging area chunk by chunk
template <int N> GCS_HD void block_chunk_A(Lane<N> &L, double *stage, int lane, bool active)
{
    using D = Dim<N>;
    static_assert(D::R1A_N <= RED_CHUNK && D::R1B_N <= RED_CHUNK && D::R1C_N <= RED_CHUNK && D::R3_N <= RED_CHUNK, "RED_CHUNK too small");
    if (!active) {
#pragma unroll
        for (int k = 0; k < D::R1A_N; ++k) stage[k * WAVE + lane] = 0.0;
        return;
    }
    block_factor<N>(L);
#pragma unroll
    for (int k = 0; k < D::NWS; ++k) stage[(D::R1_B + k) * WAVE + lane] = L.Bm[k];
#pragma unroll
    for (int k = 0; k < D::N2; ++k) stage[(D::R1_W + k) * WAVE + lane] = L.p[k];
    stage[(D::R1_W + D::N2) * WAVE + lane] = L.yy;
    stage[D::R1_GAP * WAVE + lane] = L.gap;
#pragma unroll
    for (int k = 0; k < D::NS; ++k) {
        stage[(D::R1_XII + k) * WAVE + lane] = L.X1[k];
        stage[(D::R1_XII + D::NS + k) * WAVE + lane] = L.X2[k];
    }
}
template <int N> GCS_HD void block_chunk_B(Lane<N> &L, double *stage, int lane, bool active)
{
    using D = Dim<N>;
    if (!active) {
#pragma unroll
        for (int k = 0; k < D::R1B_N; ++k) stage[k * WAVE + lane] = 0.0;
        return;
    }
    // BX columns: B * X e_c
#pragma unroll
    for (int c = 0; c < D::N2; ++c) {
        double col[D::NW], e[D::N2], bc_[D::NW];
#pragma unroll
        for (int k = 0; k < D::N2; ++k) e[k] = (k == c) ? 1.0 : 0.0;
        X_apply<N>(L, e, col);
        block_apply<N>(L, col, bc_);
#pragma unroll
        for (int i = 0; i < D::NW; ++i) { L.BX[i * D::N2 + c] = bc_[i]; stage[(i * D::N2 + c) * WAVE + lane] = bc_[i]; }
    }
    double r[D::NW];
#pragma unroll
    for (int k = 0; k < D::NW; ++k) r[k] = -L.g0[k];
    block_apply<N>(L, r, L.t0);
#pragma unroll
    for (int k = 0; k < D::NW; ++k) stage[(D::NW * D::N2 + k) * WAVE + lane] = L.t0[k];
}
template <int N> GCS_HD void block_chunk_C(Lane<N> &L, double *stage, int lane, bool active)
{
    using D = Dim<N>;
    if (!active) {
#pragma unroll
        for (int k = 0; k < D::R1C_N; ++k) stage[k * WAVE + lane] = 0.0;
        return;
    }
    // XBX = X' (B X)
#pragma unroll
    for (int c = 0; c < D::N2; ++c) {
        double col[D::NW], xt[D::N2];
#pragma unroll
        for (int i = 0; i < D::NW; ++i) col[i] = L.BX[i * D::N2 + c];
        XT_apply<N>(L, col, xt);
#pragma unroll
        for (int a = c; a < D::N2; ++a) stage[PK(a, c) * WAVE + lane] = xt[a];
    }
    double xt0[D::N2];
    XT_apply<N>(L, L.t0, xt0);
#pragma unroll
    for (int k = 0; k < D::N2; ++k) stage[(D::N2S + k) * WAVE + lane] = xt0[k];
}

// ---------------------------------------------------------------------------------------------
// cooperative segmented reduction of NV per-lane values over the block lanes of each group.
// Two halves, separated by a barrier in the caller:
//   stage_write : every lane deposits values [base, base+cnt) of its array
//   stage_sum   : lane g of a group sums value indices g, g+gsize, ... over the incoming and the
//                 outgoing block lanes separately (op: 0 sum, 1 min, 2 max for index `special`)
// ---------------------------------------------------------------------------------------------
template <int NV, int BASE, int CNT>
GCS_HD void stage_write(const double (&vals)[NV], double *stage, int lane, bool active)
{
#pragma unroll
    for (int k = 0; k < CNT; ++k) stage[k * WAVE + lane] = active ? vals[BASE + k] : 0.0;
}
template <int N>
GCS_HD void stage_sum(const Lane<N> &L, const double *stage, double *sin, double *sout, int base, int cnt,
                      int special, int op)
{
    if (L.role == IDLE || L.done) return;
    const int gsize = L.d + 1;
    for (int k = L.glane; k < cnt; k += gsize) {
        const double *row = stage + k * WAVE + L.gbase;
        const bool sp = (base + k == special);
        const double ident = (sp && op == 1) ? 1e300 : ((sp && op == 2) ? -1e300 : 0.0);
        // four loads in flight per wait: the lanes of a side are read in groups of four, out-of-range
        // slots re-read the last valid lane and contribute the identity
        auto side = [&](int lo, int hi) -> double {
            double acc = ident;
            for (int l = lo; l <= hi; l += 4) {
                double v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int idx = l + j <= hi ? l + j : hi;
                    const double x = row[idx];
                    v[j] = l + j <= hi ? x : ident;
                }
                if (sp && op == 1) acc = fmin(acc, fmin(fmin(v[0], v[1]), fmin(v[2], v[3])));
                else if (sp && op == 2) acc = fmax(acc, fmax(fmax(v[0], v[1]), fmax(v[2], v[3])));
                else acc += (v[0] + v[1]) + (v[2] + v[3]);
            }
            return acc;
        };
        sin[base + k] = side(1, L.d_in);
        sout[base + k] = side(L.d_in + 1, L.d);
    }
}

// ---------------------------------------------------------------------------------------------
// BORDER lane: convergence test, cone scaling, reduced border matrix, affine solve
// ---------------------------------------------------------------------------------------------
template <int N> struct BorderIdx {
    static constexpr int X1 = 0, X2 = N, Z1 = 2 * N, Z2 = 3 * N, YV = 4 * N, TT = 4 * N + 1;
};

// Newton solve with the stored factors.  gb: gradient (border, NB); t-sums and X't sums from LDS.
template <int N>
GCS_HD void border_solve(Lane<N> &L, double *sl, const SlotLayout<N> &SL, const double (&gb)[Dim<N>::NB],
                         const double *tin, const double *tout, const double *xtin, const double *xtout,
                         const double (&rp)[2][Dim<N>::NW], double *db, int dxoff)
{
    using D = Dim<N>;
    double rhs[D::NB];
#pragma unroll
    for (int k = 0; k < D::NB; ++k) rhs[k] = -gb[k];
#pragma unroll
    for (int c = 0; c < D::N2; ++c) rhs[c] -= xtin[c] + xtout[c];
    double u[2][D::NW];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double *Bsi = sl + SL.BSI + s * D::NWS;
        const double *BXs = sl + SL.BXS + s * D::NW * D::N2;
        const double *ts = s ? tout : tin;
        double v[D::NW];
#pragma unroll
        for (int i = 0; i < D::NW; ++i) u[s][i] = rp[s][i] - ts[i];
#pragma unroll
        for (int i = 0; i < D::NW; ++i) {
            double a = 0;
#pragma unroll
            for (int k = 0; k < D::NW; ++k) a += Bsi[PK(i, k)] * u[s][k];
            v[i] = a;
        }
#pragma unroll
        for (int c = 0; c < D::N2; ++c) {
            double a = 0;
#pragma unroll
            for (int i = 0; i < D::NW; ++i) a += BXs[i * D::N2 + c] * v[i];
            rhs[c] -= a;
        }
#pragma unroll
        for (int i = 0; i < D::NW; ++i) rhs[D::N2 + i] -= v[i];
    }
    {   // t eliminated: c0 dt + cv'(dz1 - dz2) = -gb[t]
        using BI = BorderIdx<N>;
        const double gt = gb[BI::TT] / L.soc_c0;
        double r1[D::NB1];
#pragma unroll
        for (int k = 0; k < D::NB1; ++k) r1[k] = rhs[k];
#pragma unroll
        for (int k = 0; k < N; ++k) { r1[BI::Z1 + k] += L.soc_cv[k] * gt; r1[BI::Z2 + k] -= L.soc_cv[k] * gt; }
#pragma unroll
        for (int k = 0; k < N; ++k) r1[BI::Z2 + k] += r1[BI::Z1 + k];   // rhs in the (u, z_2) variables
        chol_solve_lds<D::NB1>(sl + SL.MF, r1);
#pragma unroll
        for (int k = 0; k < N; ++k) r1[BI::Z1 + k] += r1[BI::Z2 + k];   // dz_1 = du + dz_2
#pragma unroll
        for (int k = 0; k < D::NB1; ++k) db[k] = r1[k];
        double a = -gb[BI::TT];
#pragma unroll
        for (int k = 0; k < N; ++k) a -= L.soc_cv[k] * (db[BI::Z1 + k] - db[BI::Z2 + k]);
        db[BI::TT] = a / L.soc_c0;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double *Bsi = sl + SL.BSI + s * D::NWS;
        const double *BXs = sl + SL.BXS + s * D::NW * D::N2;
        double w[D::NW];
#pragma unroll
        for (int i = 0; i < D::NW; ++i) {
            double a = db[D::N2 + i] + u[s][i];
#pragma unroll
            for (int c = 0; c < D::N2; ++c) a += BXs[i * D::N2 + c] * db[c];
            w[i] = a;
        }
#pragma unroll
        for (int i = 0; i < D::NW; ++i) {
            double a = 0;
#pragma unroll
            for (int k = 0; k < D::NW; ++k) a += Bsi[PK(i, k)] * w[k];
            sl[SL.DNU + s * D::NW + i] = a;
        }
    }
#pragma unroll
    for (int c = 0; c < D::N2; ++c) sl[dxoff + c] = db[c];
}

template <int N>
GCS_HD void border_factor_and_affine(Lane<N> &L, const WaveShared &S, const SlotLayout<N> &SL, double ipm_tol, int max_iter)
{
    using D = Dim<N>; using BI = BorderIdx<N>;
    if (L.role != BORDER || L.done) return;
    double *sl = slot_ptr<N>(S, SL, L.slot);
    const double *sin = sl + SL.SIN, *sout = sl + SL.SOUT;
    const int deg = (4 * L.m + 2) * (L.d + 1) + 1;
    // ---- complementarity and residuals ----
    double ssoc[D::Q];
    ssoc[0] = L.t;
#pragma unroll
    for (int k = 0; k < N; ++k) ssoc[1 + k] = L.p[k] - L.p[N + k];
    double gap = L.gap + sin[D::R1_GAP] + sout[D::R1_GAP];
#pragma unroll
    for (int k = 0; k < D::Q; ++k) gap += ssoc[k] * L.lsoc[k];
    const double mu = gap / deg;
    L.mu = mu;
#if !defined(__HIPCC__) && defined(GCS_EMU_TRACE)
    if (getenv("GCS_EMU_TRACE")) fprintf(stderr, "v %d it %d mu %.17g\n", L.v, L.iters, mu);
#endif
    double rp[2][D::NW];
    double rpmax = 0;
#pragma unroll
    for (int k = 0; k < D::NW; ++k) {
        const double zeta = (k < D::N2) ? L.p[k] : L.yy;
        rp[0][k] = zeta - sin[D::R1_W + k];
        rp[1][k] = zeta - sout[D::R1_W + k];
        rpmax = fmax(rpmax, fmax(fabs(rp[0][k]), fabs(rp[1][k])));
    }
    // stop on the barrier parameter alone (see oracle/gcs_oracle.c): insensitive to solve round-off
    (void)rpmax;
    // a vanishing step (stalled) means the linear algebra has run out of precision: accept the point if
    // the barrier parameter is within 1e3 of the target
    const bool conv = mu <= ipm_tol || (L.stalled && mu <= 1e3 * ipm_tol);
    bool stop = conv;
    int status = conv ? 0 : -1;
    if (!(mu > 0.0)) { stop = true; status = -3; }   // a slack left the cone (NaN from pass_A)
    if (!stop && L.iters >= max_iter) { stop = true; status = -1; }
    // ---- cone scaling ----
    double wb[D::Q], eta = 1.0;
#pragma unroll
    for (int k = 0; k < D::Q; ++k) wb[k] = 0.0;
    if (!stop) {
        if (!soc_scaling<D::Q>(ssoc, L.lsoc, L.W, L.Wi, wb, eta)) { stop = true; status = mu <= 1e3 * ipm_tol ? 0 : -4; }
    }
    if (stop) {
        L.done = 1; L.status = status;
        sl[SL.SC + 2] = 1.0;
        return;
    }
#pragma unroll
    for (int i = 0; i < D::Q; ++i) {
#pragma unroll
        for (int j = 0; j < D::Q; ++j) {
            double a = 0;
#pragma unroll
            for (int k = 0; k < D::Q; ++k) a += L.Wi[i * D::Q + k] * L.Wi[k * D::Q + j];
            L.W2[i * D::Q + j] = a;
        }
        double a = 0;
#pragma unroll
        for (int k = 0; k < D::Q; ++k) a += L.W[i * D::Q + k] * L.lsoc[k];
        L.lt[i] = a;
    }
    // ---- per side: factor B_s, inverse, Y_s = Bs^{-1} BXs (all kept in the LDS slot) ----
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double *sum = s ? sout : sin;
        double Bs[D::NWS], Bsi[D::NWS];
#pragma unroll
        for (int k = 0; k < D::NWS; ++k) Bs[k] = sum[D::R1_B + k];
        chol_packed<D::NW>(Bs);
        chol_inverse_packed<D::NW>(Bs, Bsi);
#pragma unroll
        for (int k = 0; k < D::NWS; ++k) sl[SL.BSI + s * D::NWS + k] = Bsi[k];
#pragma unroll
        for (int i = 0; i < D::NW; ++i)
#pragma unroll
            for (int c = 0; c < D::N2; ++c) {
                double a = 0;
#pragma unroll
                for (int k = 0; k < D::NW; ++k) a += Bsi[PK(i, k)] * sum[D::R1_BX + k * D::N2 + c];
                sl[SL.YS + s * D::NW * D::N2 + i * D::N2 + c] = a;
                sl[SL.BXS + s * D::NW * D::N2 + i * D::N2 + c] = sum[D::R1_BX + i * D::N2 + c];
            }
    }
    // ---- reduced border matrix M (packed lower, order x1 x2 z1 z2 yv), assembled entry by entry in LDS ----
    double M[D::NBS];      // in registers: the LDS-resident variant costs ~3x (one LDS instruction per operand)
    const double *Y0 = sl + SL.YS, *Y1 = sl + SL.YS + D::NW * D::N2;
    const double *BX0 = sl + SL.BXS, *BX1 = sl + SL.BXS + D::NW * D::N2;
    const double *Bi0 = sl + SL.BSI, *Bi1 = sl + SL.BSI + D::NWS;
    // x-x block
#pragma unroll
    for (int a = 0; a < D::N2; ++a)
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            double v = (a == c ? REG_DELTA : 0.0) - (sin[D::R1_XBX + PK(a, c)] + sout[D::R1_XBX + PK(a, c)]);
            if (a < N && c < N) v += L.X1[PK(a, c)] + sin[D::R1_XII + PK(a, c)] + sout[D::R1_XII + PK(a, c)];
            if (a >= N && c >= N) v += L.X2[PK(a - N, c - N)] + sin[D::R1_XII + D::NS + PK(a - N, c - N)] + sout[D::R1_XII + D::NS + PK(a - N, c - N)];
#pragma unroll
            for (int k = 0; k < D::NW; ++k) v += BX0[k * D::N2 + a] * Y0[k * D::N2 + c] + BX1[k * D::N2 + a] * Y1[k * D::N2 + c];
            M[PK(a, c)] = v;
        }
    // zeta-x block (rows z1, z2, yv ; columns x1, x2)
#pragma unroll
    for (int i = 0; i < D::NW; ++i)
#pragma unroll
        for (int c = 0; c < D::N2; ++c) {
            double v = Y0[i * D::N2 + c] + Y1[i * D::N2 + c];
            if (i < N && c < N) v -= L.X1[PK(i, c)];
            if (i >= N && i < D::N2 && c >= N) v -= L.X2[PK(i - N, c - N)];
            if (i == D::N2) v += (c < N) ? L.xy1[c] : L.xy2[c - N];
            M[PK(D::N2 + i, c)] = v;
        }
    // zeta-zeta block
#pragma unroll
    for (int i = 0; i < D::NW; ++i)
#pragma unroll
        for (int k = 0; k <= i; ++k) {
            double v = Bi0[PK(i, k)] + Bi1[PK(i, k)];
            if (i < N) v += L.K1[PK(i, k)];
            if (i >= N && i < D::N2 && k >= N) v += L.K2[PK(i - N, k - N)];
            if (i == D::N2) v += (k < N) ? L.k1y[k] : (k < D::N2 ? L.k2y[k - N] : L.kyy);
            M[PK(D::N2 + i, D::N2 + k)] = v;
        }
    // Cone block.  With W^{-2} = eta^{-2}(2 v v' - J), v = (wb0, -wb1), eliminating t first leaves on
    // u = z_1 - z_2 the Schur complement Su = eta^{-2}(I - 2 wb1 wb1'/(2 wb0^2 - 1)), formed from this
    // closed form: a numerical pivot on t cancels catastrophically once the cone is active.
    double Su[N * N];
    {
        const double ie2 = rcp(eta * eta), g2 = 2.0 * rcp(2.0 * wb[0] * wb[0] - 1.0);
        L.soc_c0 = ie2 * (2.0 * wb[0] * wb[0] - 1.0);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            L.soc_cv[k] = -ie2 * 2.0 * wb[0] * wb[1 + k];
#pragma unroll
            for (int l = 0; l < N; ++l) Su[k * N + l] = ie2 * ((k == l ? 1.0 : 0.0) - g2 * wb[1 + k] * wb[1 + l]);
        }
    }
    // Change of variables (u, z_2) = (z_1 - z_2, z_2): the cone term then sits on u alone.  In (z_1, z_2)
    // it enters as [Su -Su; -Su Su] and, with the cone inactive (t -> 0), Su grows like 1/mu, so that
    // eliminating z_1 before z_2 would cancel K_2 + Su - Su (K_1 + Su)^{-1} Su.
    {   // in place: first the (z2', z2') block, which needs the untouched cross entries, then the rest
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int l = 0; l <= k; ++l)
                M[PK(BI::Z2 + k, BI::Z2 + l)] += M[PK(BI::Z1 + k, BI::Z2 + l)] + M[PK(BI::Z2 + k, BI::Z1 + l)] + M[PK(BI::Z1 + k, BI::Z1 + l)];
#pragma unroll
        for (int i = 0; i < D::NB1; ++i) {
            if (i >= BI::Z2 && i < BI::Z2 + N) continue;
#pragma unroll
            for (int k = 0; k < N; ++k) M[PK(i, BI::Z2 + k)] += M[PK(i, BI::Z1 + k)];
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int l = 0; l <= k; ++l) M[PK(BI::Z1 + k, BI::Z1 + l)] += Su[k * N + l];
        chol_packed<D::NB1>(M);
#pragma unroll
        for (int k = 0; k < D::NBS; ++k) sl[SL.MF + k] = M[k];
    }
    // ---- affine direction (kappa = 0) ----
    double gb[D::NB];
#pragma unroll
    for (int k = 0; k < D::N2; ++k) gb[k] = REG_DELTA * L.x[k];
#pragma unroll
    for (int k = 0; k < D::NW; ++k) gb[D::N2 + k] = L.g0[k];
    gb[BI::TT] = 1.0;
    border_solve<N>(L, sl, SL, gb, sin + D::R1_T, sout + D::R1_T, sin + D::R1_XT, sout + D::R1_XT, rp, L.dba, SL.DXA);
}

// ---------------------------------------------------------------------------------------------
// direction of a lane's own unknowns from the border solution in LDS
// ---------------------------------------------------------------------------------------------
template <int N>
GCS_HD void lane_direction(Lane<N> &L, const double *sl, const SlotLayout<N> &SL, bool affine, double (&dp)[2 * N + 1])
{
    using D = Dim<N>;
    if (L.role == BLOCK) {
        const double *dnu = sl + SL.DNU + (L.out ? D::NW : 0);
        double xd[D::NW], r[D::NW];
        X_apply<N>(L, sl + (affine ? SL.DXA : SL.DX), xd);
#pragma unroll
        for (int k = 0; k < D::NW; ++k) r[k] = -(L.g0[k] + (affine ? 0.0 : L.gk[k])) + dnu[k] - xd[k];
        block_apply<N>(L, r, dp);
    } else {
        const double *db = affine ? L.dba : L.db;
#pragma unroll
        for (int k = 0; k < D::NW; ++k) dp[k] = db[D::N2 + k];
    }
}

// PASS B (after the affine solve): step bound, mu_aff sums, and the two vectors G'(1/s), G'(corr/s)
template <int N, int I>
GCS_HD void pass_B_half(Lane<N> &L, const double *A, const double *bc, const double *x, const double *dxa,
                        const double *la, const double *lb, int MM, int lane)
{
    rows_half<N, I>(L, A, bc, x, [&](int j, const double(&a)[N], double b, double sa, double sb) {
        const double l_a = la[(I * MM + j) * WAVE + lane], l_b = lb[(I * MM + j) * WAVE + lane];
        double dsa, dsb;
        row_dir<N, I>(a, b, L.dpa, dxa, dsa, dsb);
        const double ia = rcp(sa), ib = rcp(sb);
        const double qsa = dsa * ia, qsb = dsb * ib;                 // ds / s
        const double dla = -l_a - l_a * qsa, dlb = -l_b - l_b * qsb; // kappa = 0:  dl / l = -1 - ds / s
        // largest -ds/s and -dl/l over the rows; the step bound is its reciprocal
        L.amax = fmax(L.amax, fmax(fmax(-qsa, 1.0 + qsa), fmax(-qsb, 1.0 + qsb)));
        L.c1 += sa * dla + l_a * dsa + sb * dlb + l_b * dsb;
        L.c2 += dsa * dla + dsb * dlb;
        const double qa = dsa * dla * ia, qb = dsb * dlb * ib;
        // G rows: a-type (+a on p_i, -b on y), b-type (-a on p_i, +b on y, +a on x_i)
#pragma unroll
        for (int k = 0; k < N; ++k) {
            L.P1[I * N + k] += a[k] * (ia - ib);
            L.P2[I * N + k] += a[k] * (qa - qb);
            L.P1[2 * N + 1 + I * N + k] += a[k] * ib;
            L.P2[2 * N + 1 + I * N + k] += a[k] * qb;
        }
        L.P1[2 * N] += b * (ib - ia);
        L.P2[2 * N] += b * (qb - qa);
    });
}

template <int N>
GCS_HD void pass_B(Lane<N> &L, const WaveShared &S, const SlotLayout<N> &SL, int lane, int par)
{
    using D = Dim<N>;
    if (L.role == IDLE || L.done) return;
    double *sl = slot_ptr<N>(S, SL, L.slot);
    if (sl[SL.SC + 2] != 0.0) { L.done = 1; return; }
    lane_direction<N>(L, sl, SL, true, L.dpa);
    L.amax = 0.0; L.c1 = 0; L.c2 = 0;   // amax holds the largest ratio until the end of the pass
#pragma unroll
    for (int k = 0; k < 2 * D::N2 + 1; ++k) { L.P1[k] = 0; L.P2[k] = 0; }
    const double *dxa = sl + SL.DXA, *x = sl + SL.X + par * D::N2;
    pass_B_half<N, 0>(L, sl + SL.A, sl + SL.B, x, dxa, S.lamA, S.lamB, S.MM, lane);
    pass_B_half<N, 1>(L, sl + SL.A, sl + SL.B, x, dxa, S.lamA, S.lamB, S.MM, lane);
    {   // bounds: s5 = yy (G = -1), s6 = 1 - yy (G = +1)
        const double s5 = L.yy, s6 = 1.0 - L.yy, ds5 = L.dpa[2 * N], ds6 = -L.dpa[2 * N];
        const double i5 = rcp(s5), i6 = rcp(s6), q5 = ds5 * i5, q6 = ds6 * i6;
        const double dl5 = -L.l5 - L.l5 * q5, dl6 = -L.l6 - L.l6 * q6;
        L.amax = fmax(L.amax, fmax(fmax(-q5, 1.0 + q5), fmax(-q6, 1.0 + q6)));
        L.c1 += s5 * dl5 + L.l5 * ds5 + s6 * dl6 + L.l6 * ds6;
        L.c2 += ds5 * dl5 + ds6 * dl6;
        L.P1[2 * N] += -i5 + i6;
        L.P2[2 * N] += -(ds5 * dl5) * i5 + (ds6 * dl6) * i6;
    }
    L.amax = L.amax > 0.0 ? rcp(L.amax) : 1e300;
    if (L.role == BORDER) {
        using BI = BorderIdx<N>;
        double ssoc[D::Q];
        ssoc[0] = L.t;
        L.dssa[0] = L.dba[BI::TT];
#pragma unroll
        for (int k = 0; k < N; ++k) { ssoc[1 + k] = L.p[k] - L.p[N + k]; L.dssa[1 + k] = L.dpa[k] - L.dpa[N + k]; }
#pragma unroll
        for (int i = 0; i < D::Q; ++i) {
            double a = -L.lsoc[i];
#pragma unroll
            for (int k = 0; k < D::Q; ++k) a -= L.W2[i * D::Q + k] * L.dssa[k];
            L.dlsa[i] = a;
        }
        L.amax = fmin(L.amax, fmin(soc_max_step<D::Q>(ssoc, L.dssa), soc_max_step<D::Q>(L.lsoc, L.dlsa)));
#pragma unroll
        for (int k = 0; k < D::Q; ++k) { L.c1 += ssoc[k] * L.dlsa[k] + L.lsoc[k] * L.dssa[k]; L.c2 += L.dssa[k] * L.dlsa[k]; }
    }
}

// BORDER: centring parameter from the affine step
template <int N>
GCS_HD void border_sigma(Lane<N> &L, const WaveShared &S, const SlotLayout<N> &SL)
{
    using D = Dim<N>;
    if (L.role != BORDER || L.done) return;
    double *sl = slot_ptr<N>(S, SL, L.slot);
    const double *sin = sl + SL.SIN, *sout = sl + SL.SOUT;
    const int deg = (4 * L.m + 2) * (L.d + 1) + 1;
    const double amax = fmin(L.amax, fmin(sin[D::R2_BASE], sout[D::R2_BASE]));
    const double c1 = L.c1 + sin[D::R2_BASE + 1] + sout[D::R2_BASE + 1], c2 = L.c2 + sin[D::R2_BASE + 2] + sout[D::R2_BASE + 2];
    const double al = fmin(1.0, amax);
    const double gap = L.mu * deg;
    const double mu_aff = (gap + al * c1 + al * al * c2) / deg;
    double sig = mu_aff / L.mu;
    sig = sig < 0 ? 0 : (sig > 1 ? 1 : sig);
    sig = sig * sig * sig;
    const double sm = sig * L.mu;
    sl[SL.SC + 1] = sm;
    // cone part of kappa: sigma mu s^{-1} - W^{-1}( lt \ ((W^{-1} ds_a) o (W dl_a)) )
    double ssoc[D::Q], a1[D::Q], a2[D::Q], pr[D::Q], qv[D::Q];
    ssoc[0] = L.t;
#pragma unroll
    for (int k = 0; k < N; ++k) ssoc[1 + k] = L.p[k] - L.p[N + k];
#pragma unroll
    for (int i = 0; i < D::Q; ++i) {
        double u1 = 0, u2 = 0;
#pragma unroll
        for (int k = 0; k < D::Q; ++k) { u1 += L.Wi[i * D::Q + k] * L.dssa[k]; u2 += L.W[i * D::Q + k] * L.dlsa[k]; }
        a1[i] = u1; a2[i] = u2;
    }
    {   // pr = a1 o a2
        double dsum = 0;
#pragma unroll
        for (int k = 0; k < D::Q; ++k) dsum += a1[k] * a2[k];
        pr[0] = dsum;
#pragma unroll
        for (int k = 1; k < D::Q; ++k) pr[k] = a1[0] * a2[k] + a2[0] * a1[k];
    }
    {   // lt o qv = pr
        const double det = soc_det<D::Q>(L.lt);
        double ld1 = 0;
#pragma unroll
        for (int k = 1; k < D::Q; ++k) ld1 += L.lt[k] * pr[k];
        qv[0] = (L.lt[0] * pr[0] - ld1) / det;
#pragma unroll
        for (int k = 1; k < D::Q; ++k) qv[k] = (pr[k] - qv[0] * L.lt[k]) / L.lt[0];
    }
    const double dets = soc_det<D::Q>(ssoc);
#pragma unroll
    for (int i = 0; i < D::Q; ++i) {
        double a = 0;
#pragma unroll
        for (int k = 0; k < D::Q; ++k) a += L.Wi[i * D::Q + k] * qv[k];
        L.ksoc[i] = sm * (i == 0 ? ssoc[0] : -ssoc[i]) / dets - a;
    }
}

// all lanes: G'kappa = sigma mu P1 - P2 ; block lanes stage t1 = B(-(g0+gk)), X't1, gk_x
template <int N>
GCS_HD void corrector_rhs(Lane<N> &L, const WaveShared &S, const SlotLayout<N> &SL, double (&vals)[Dim<N>::R3_N])
{
    using D = Dim<N>;
#pragma unroll
    for (int k = 0; k < D::R3_N; ++k) vals[k] = 0;
    if (L.role == IDLE || L.done) return;
    double *sl = slot_ptr<N>(S, SL, L.slot);
    const double sm = sl[SL.SC + 1];
#pragma unroll
    for (int k = 0; k < 2 * D::N2 + 1; ++k) L.gk[k] = sm * L.P1[k] - L.P2[k];
    if (L.role == BLOCK) {
        double r[D::NW], t1[D::NW], xt[D::N2];
#pragma unroll
        for (int k = 0; k < D::NW; ++k) r[k] = -(L.g0[k] + L.gk[k]);
        block_apply<N>(L, r, t1);
        XT_apply<N>(L, t1, xt);
#pragma unroll
        for (int k = 0; k < D::NW; ++k) vals[D::R3_T + k] = t1[k];
#pragma unroll
        for (int k = 0; k < D::N2; ++k) { vals[D::R3_XT + k] = xt[k]; vals[D::R3_GX + k] = L.gk[D::NW + k]; }
    }
}

template <int N>
GCS_HD void border_corrector_solve(Lane<N> &L, const WaveShared &S, const SlotLayout<N> &SL)
{
    using D = Dim<N>; using BI = BorderIdx<N>;
    if (L.role != BORDER || L.done) return;
    double *sl = slot_ptr<N>(S, SL, L.slot);
    const double *sin = sl + SL.SIN, *sout = sl + SL.SOUT;
    double gb[D::NB];
#pragma unroll
    for (int k = 0; k < D::N2; ++k) gb[k] = REG_DELTA * L.x[k] + L.gk[D::NW + k] + sin[D::R3_BASE + D::R3_GX + k] + sout[D::R3_BASE + D::R3_GX + k];
#pragma unroll
    for (int k = 0; k < D::NW; ++k) gb[D::N2 + k] = L.g0[k] + L.gk[k];
#pragma unroll
    for (int k = 0; k < N; ++k) { gb[BI::Z1 + k] -= L.ksoc[1 + k]; gb[BI::Z2 + k] += L.ksoc[1 + k]; }
    gb[BI::TT] = 1.0 - L.ksoc[0];
    double rp[2][D::NW];
#pragma unroll
    for (int k = 0; k < D::NW; ++k) {
        const double zeta = (k < D::N2) ? L.p[k] : L.yy;
        rp[0][k] = zeta - sin[D::R1_W + k];
        rp[1][k] = zeta - sout[D::R1_W + k];
    }
    border_solve<N>(L, sl, SL, gb, sin + D::R3_BASE + D::R3_T, sout + D::R3_BASE + D::R3_T,
                    sin + D::R3_BASE + D::R3_XT, sout + D::R3_BASE + D::R3_XT, rp, L.db, SL.DX);
}

// PASS D / E: final direction: step bound (APPLY = false) or dual update by alpha (APPLY = true).
// kappa of a row is rebuilt from the affine direction: kappa = (sigma mu - ds_a dl_a) / s.
template <int N, int I, bool APPLY>
GCS_HD void pass_DE_half(Lane<N> &L, const double *A, const double *bc, const double *x, const double *dxa,
                         const double *dx, double *la, double *lb, int MM, int lane, double sm, double alpha)
{
    rows_half<N, I>(L, A, bc, x, [&](int j, const double(&a)[N], double b, double sa, double sb) {
        double &ra = la[(I * MM + j) * WAVE + lane];
        double &rb = lb[(I * MM + j) * WAVE + lane];
        const double isa = rcp(sa), isb = rcp(sb);
        const double l_a = ra, l_b = rb, Da = l_a * isa, Db = l_b * isb;
        double dsa0, dsb0, dsa, dsb;
        row_dir<N, I>(a, b, L.dpa, dxa, dsa0, dsb0);
        row_dir<N, I>(a, b, L.dp, dx, dsa, dsb);
        const double ka = (sm - dsa0 * (-l_a - Da * dsa0)) * isa, kb = (sm - dsb0 * (-l_b - Db * dsb0)) * isb;
        const double dla = ka - l_a - Da * dsa, dlb = kb - l_b - Db * dsb;
        if (APPLY) {
            ra = l_a + alpha * dla;
            rb = l_b + alpha * dlb;
        } else {
            // largest -ds/s and -dl/l; the step bound is the reciprocal (taken once, in pass_DE)
            L.amax = fmax(L.amax, fmax(fmax(-dsa * isa, -dla * rcp(l_a)), fmax(-dsb * isb, -dlb * rcp(l_b))));
        }
    });
}

template <int N, bool APPLY>
GCS_HD void pass_DE(Lane<N> &L, const WaveShared &S, const SlotLayout<N> &SL, int lane, int par)
{
    using D = Dim<N>; using BI = BorderIdx<N>;
    if (L.role == IDLE || L.done) return;
    double *sl = slot_ptr<N>(S, SL, L.slot);
    const double sm = sl[SL.SC + 1];
    const double alpha = APPLY ? sl[SL.SC + 0] : 0.0;
    if (!APPLY) {
        if (L.role == BORDER) {
#pragma unroll
            for (int k = 0; k < D::NW; ++k) L.dp[k] = L.db[D::N2 + k];
        } else {
            lane_direction<N>(L, sl, SL, false, L.dp);
        }
        L.amax = 0.0;   // largest ratio until converted below
    }
    const double *x = sl + SL.X + par * D::N2;
    pass_DE_half<N, 0, APPLY>(L, sl + SL.A, sl + SL.B, x, sl + SL.DXA, sl + SL.DX, S.lamA, S.lamB, S.MM, lane, sm, alpha);
    pass_DE_half<N, 1, APPLY>(L, sl + SL.A, sl + SL.B, x, sl + SL.DXA, sl + SL.DX, S.lamA, S.lamB, S.MM, lane, sm, alpha);
    {
        const double s5 = L.yy, s6 = 1.0 - L.yy;
        const double i5 = rcp(s5), i6 = rcp(s6);
        const double D5 = L.l5 * i5, D6 = L.l6 * i6;
        const double d50 = L.dpa[2 * N], d60 = -L.dpa[2 * N], d5 = L.dp[2 * N], d6 = -L.dp[2 * N];
        const double k5 = (sm - d50 * (-L.l5 - D5 * d50)) * i5, k6 = (sm - d60 * (-L.l6 - D6 * d60)) * i6;
        const double dl5 = k5 - L.l5 - D5 * d5, dl6 = k6 - L.l6 - D6 * d6;
        if (APPLY) {
            L.l5 += alpha * dl5; L.l6 += alpha * dl6;
        } else {
            L.amax = fmax(L.amax, fmax(fmax(-d5 * i5, -dl5 * rcp(L.l5)), fmax(-d6 * i6, -dl6 * rcp(L.l6))));
            L.amax = L.amax > 0.0 ? rcp(L.amax) : 1e300;
        }
    }
    if (L.role == BORDER && !APPLY) {
        double ssoc[D::Q];
        ssoc[0] = L.t;
        L.dss[0] = L.db[BI::TT];
#pragma unroll
        for (int k = 0; k < N; ++k) { ssoc[1 + k] = L.p[k] - L.p[N + k]; L.dss[1 + k] = L.dp[k] - L.dp[N + k]; }
#pragma unroll
        for (int i = 0; i < D::Q; ++i) {
            double a = L.ksoc[i] - L.lsoc[i];
#pragma unroll
            for (int k = 0; k < D::Q; ++k) a -= L.W2[i * D::Q + k] * L.dss[k];
            L.dls[i] = a;
        }
        L.amax = fmin(L.amax, fmin(soc_max_step<D::Q>(ssoc, L.dss), soc_max_step<D::Q>(L.lsoc, L.dls)));
    }
    if (APPLY) {
        // primal update of the lane's own unknowns
#pragma unroll
        for (int k = 0; k < D::N2; ++k) L.p[k] += alpha * L.dp[k];
        L.yy += alpha * L.dp[2 * N];
        if (L.role == BORDER) {
            double *xn = sl + SL.X + (par ^ 1) * D::N2;
#pragma unroll
            for (int k = 0; k < D::N2; ++k) { L.x[k] += alpha * L.db[k]; xn[k] = L.x[k]; }
            L.t += alpha * L.db[BI::TT];
#pragma unroll
            for (int k = 0; k < D::Q; ++k) L.lsoc[k] += alpha * L.dls[k];
#pragma unroll
            for (int k = 0; k < 2 * D::NW; ++k) sl[SL.NU + k] += alpha * sl[SL.DNU + k];
            L.iters += 1;
        }
    }
}

// BORDER: step length with the cone guard
template <int N>
GCS_HD void border_alpha(Lane<N> &L, const WaveShared &S, const SlotLayout<N> &SL)
{
    using D = Dim<N>;
    if (L.role != BORDER || L.done) return;
    double *sl = slot_ptr<N>(S, SL, L.slot);
    const double amax = fmin(L.amax, fmin(sl[SL.SIN + D::R4_BASE], sl[SL.SOUT + D::R4_BASE]));
    double al = fmin(1.0, 0.99 * amax);
    double ssoc[D::Q];
    ssoc[0] = L.t;
#pragma unroll
    for (int k = 0; k < N; ++k) ssoc[1 + k] = L.p[k] - L.p[N + k];
    for (int tries = 0; tries < 40; ++tries) {
        double s2[D::Q], l2[D::Q];
#pragma unroll
        for (int k = 0; k < D::Q; ++k) { s2[k] = ssoc[k] + al * L.dss[k]; l2[k] = L.lsoc[k] + al * L.dls[k]; }
        if (soc_interior<D::Q>(s2) && soc_interior<D::Q>(l2)) break;
        al *= 0.7;
    }
#if !defined(__HIPCC__) && defined(GCS_EMU_TRACE)
    if (getenv("GCS_EMU_TRACE")) fprintf(stderr, "v %d it %d alpha %.6g own %.6g in %.6g out %.6g sm %.3g\n", L.v, L.iters, al, L.amax, sl[SL.SIN + D::R4_BASE], sl[SL.SOUT + D::R4_BASE], sl[SL.SC+1]);
#endif
    L.stalled = al < 1e-3;
    sl[SL.SC + 0] = al;
}


// ---------------------------------------------------------------------------------------------
// global-memory interface of the vertex step and the wave-level driver
// ---------------------------------------------------------------------------------------------
struct ControlView {   // the fields of gcsadmm_control_block the kernels read
    double rho, mu_scale;
    int status;
};

template <class T> struct VertexArgs {
    int n_waves;
    const int *wave_slot_ptr;   // [n_waves+1] into wave_vtx
    const int *wave_vtx;        // vertex ids, grouped per wave
    const int *inc_ptr;         // [V+1]
    const int *deg_in;          // [V]
    const int *inc_edge;        // [NI_owned]
    const int *poly_ptr;        // [V+1]
    const double *poly_A;       // [sum m][n]
    const double *poly_bc;      // [sum m] centred: b - A c
    const double *center;       // [V][n]
    int E, NI, MM;
    const T *zedge, *mu;
    T *copy;
    double *xv, *zv, *yv;
    int *counters;              // [0] inner failures, [1] inner iterations
    double eps_edge, ipm_tol;
    int ipm_max_iter;
};

template <int N, class T>
GCS_HD void phase_setup(Lane<N> &L, int lane, int wave, const VertexArgs<T> &a, const WaveShared &S, const SlotLayout<N> &SL)
{
    using D = Dim<N>;
    L.role = IDLE; L.slot = 0; L.v = 0; L.d = 0; L.d_in = 0; L.glane = 0; L.gbase = 0; L.m = 0;
    L.inc = 0; L.edge = 0; L.out = 0; L.done = 0; L.iters = 0; L.status = -1; L.stalled = 0; L.bad = 0;
    const int s0 = a.wave_slot_ptr[wave], s1 = a.wave_slot_ptr[wave + 1];
    int base = 0;
    for (int s = s0; s < s1; ++s) {
        const int v = a.wave_vtx[s];
        const int lo = a.inc_ptr[v], d = a.inc_ptr[v + 1] - lo;
        if (lane >= base && lane <= base + d) {
            L.slot = s - s0; L.v = v; L.d = d; L.d_in = a.deg_in[v]; L.gbase = base; L.glane = lane - base;
            L.m = a.poly_ptr[v + 1] - a.poly_ptr[v];
            if (L.glane == 0) L.role = BORDER;
            else {
                L.role = BLOCK;
                L.inc = lo + L.glane - 1;
                L.edge = a.inc_edge[L.inc];
                L.out = (L.glane - 1) >= L.d_in;
            }
        }
        base += d + 1;
    }
    {
        double *bp = slot_ptr<N>(S, SL, L.slot) + SL.BORD;
        L.lsoc = bp; bp += D::Q;
        L.x = bp; bp += D::N2;
        L.W = bp; bp += D::Q * D::Q;
        L.Wi = bp; bp += D::Q * D::Q;
        L.W2 = bp; bp += D::Q * D::Q;
        L.lt = bp; bp += D::Q;
        L.soc_cv = bp; bp += N;
        L.dba = bp; bp += D::NB;
        L.db = bp; bp += D::NB;
        L.dssa = bp; bp += D::Q;
        L.dlsa = bp; bp += D::Q;
        L.dss = bp; bp += D::Q;
        L.dls = bp; bp += D::Q;
        L.ksoc = bp;
    }
    if (L.role == BORDER) {
        double *sl = slot_ptr<N>(S, SL, L.slot);
        const int p0 = a.poly_ptr[L.v];
        for (int j = 0; j < L.m; ++j) {
#pragma unroll
            for (int k = 0; k < N; ++k) sl[SL.A + j * N + k] = a.poly_A[(size_t)(p0 + j) * N + k];
            sl[SL.B + j] = a.poly_bc[p0 + j];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) sl[SL.CEN + k] = a.center[(size_t)L.v * N + k];
#pragma unroll
        for (int k = 0; k < 2 * D::N2; ++k) sl[SL.X + k] = 0.0;
#pragma unroll
        for (int k = 0; k < 2 * D::NW; ++k) sl[SL.NU + k] = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sl[SL.SC + k] = 0.0;
    }
}

template <int N, class T>
GCS_HD void phase_load(Lane<N> &L, int lane, const VertexArgs<T> &a, const WaveShared &S, const SlotLayout<N> &SL,
                       double rho, double mu_scale)
{
    using D = Dim<N>;
    L.scale0 = 0.0;
    if (L.role == BLOCK) {
        const double *cen = slot_ptr<N>(S, SL, L.slot) + SL.CEN;
        double Tw[D::NW];
#pragma unroll
        for (int w = 0; w < D::NW; ++w)
            Tw[w] = (double)a.zedge[(size_t)w * a.E + L.edge] - mu_scale * (double)a.mu[(size_t)w * a.NI + L.inc];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            L.T1[k] = L.out ? Tw[k] : Tw[N + k];
            L.T2[k] = L.out ? Tw[N + k] : 0.0;
            L.Tfree[k] = Tw[k];
        }
        L.Ty = Tw[2 * N];
#pragma unroll
        for (int k = 0; k < D::N2; ++k) L.p[k] = 0.0;
        L.yy = 0.5 / (double)(L.out ? (L.d - L.d_in) : L.d_in);
        double sc = 1.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            sc = fmax(sc, 1.0 + fabs(rho * (L.yy * cen[k] - L.T1[k])));
            if (L.out) sc = fmax(sc, 1.0 + fabs(rho * (L.yy * cen[k] - L.T2[k])));
        }
        sc = fmax(sc, 1.0 + fabs(rho * (L.yy - L.Ty)));
        L.scale0 = sc;
    } else if (L.role == BORDER) {
#pragma unroll
        for (int k = 0; k < D::N2; ++k) { L.p[k] = 0.0; L.x[k] = 0.0; }
        L.yy = 0.5; L.t = 1.0;
        L.lsoc[0] = 1.0;
#pragma unroll
        for (int k = 1; k < D::Q; ++k) L.lsoc[k] = 0.0;
        L.scale = 1.0;
    }
}

// the wave-level program.  EX provides each(f) = run f(L, lane) for every lane, then a barrier,
// all(pred) = wave-wide vote, count(counters, fails, iters) = accumulate statistics.
template <int N, class T, class EX>
GCS_HD void run_vertex_program(EX &ex, int wave, const VertexArgs<T> &a, const WaveShared &S, double rho, double mu_scale)
{
    using D = Dim<N>;
    const SlotLayout<N> SL(S.MM);
    ex.each([&](Lane<N> &L, int lane) { phase_setup<N, T>(L, lane, wave, a, S, SL); });
    ex.each([&](Lane<N> &L, int lane) { phase_load<N, T>(L, lane, a, S, SL, rho, mu_scale); });
    // scale of the objective gradient at the start: max over the block lanes of each group
    ex.each([&](Lane<N> &L, int lane) { S.stage[lane] = L.role == BLOCK ? L.scale0 : 0.0; });
    ex.each([&](Lane<N> &L, int) {
        if (L.role == IDLE) return;
        double *sl = slot_ptr<N>(S, SL, L.slot);
        stage_sum<N>(L, S.stage, sl + SL.SIN, sl + SL.SOUT, D::R0_BASE, 1, D::R0_BASE, 2);
    });
    ex.each([&](Lane<N> &L, int) {
        if (L.role != BORDER) return;
        double *sl = slot_ptr<N>(S, SL, L.slot);
        L.scale = fmax(1.0, fmax(sl[SL.SIN + D::R0_BASE], sl[SL.SOUT + D::R0_BASE]));
    });
    for (int it = 0;; ++it) {
        const int par = it & 1;
        // ---- round 1: rows, block factorisation, sums (three chunks, each computed just before staging) ----
        ex.each([&](Lane<N> &L, int lane) {
            pass_A<N>(L, S, SL, lane, it == 0, par, rho, a.eps_edge);
            block_chunk_A<N>(L, S.stage, lane, L.role == BLOCK && !L.done);
        });
        ex.each([&](Lane<N> &L, int) { if (L.role != IDLE) { double *sl = slot_ptr<N>(S, SL, L.slot); stage_sum<N>(L, S.stage, sl + SL.SIN, sl + SL.SOUT, 0, D::R1A_N, -1, 0); } });
        ex.each([&](Lane<N> &L, int lane) { block_chunk_B<N>(L, S.stage, lane, L.role == BLOCK && !L.done); });
        ex.each([&](Lane<N> &L, int) { if (L.role != IDLE) { double *sl = slot_ptr<N>(S, SL, L.slot); stage_sum<N>(L, S.stage, sl + SL.SIN, sl + SL.SOUT, D::R1_BX, D::R1B_N, -1, 0); } });
        ex.each([&](Lane<N> &L, int lane) { block_chunk_C<N>(L, S.stage, lane, L.role == BLOCK && !L.done); });
        ex.each([&](Lane<N> &L, int) { if (L.role != IDLE) { double *sl = slot_ptr<N>(S, SL, L.slot); stage_sum<N>(L, S.stage, sl + SL.SIN, sl + SL.SOUT, D::R1_XBX, D::R1C_N, -1, 0); } });
        // ---- border: convergence, factorisation, affine solve ----
        ex.each([&](Lane<N> &L, int) { border_factor_and_affine<N>(L, S, SL, a.ipm_tol, a.ipm_max_iter); });
        if (ex.all([&](Lane<N> &L) { return L.role == IDLE || slot_ptr<N>(S, SL, L.slot)[SL.SC + 2] != 0.0; })) break;
        // ---- affine step statistics ----
        ex.each([&](Lane<N> &L, int lane) {
            pass_B<N>(L, S, SL, lane, par);
            const bool act = L.role == BLOCK && !L.done;
            S.stage[0 * WAVE + lane] = act ? L.amax : 1e300;
            S.stage[1 * WAVE + lane] = act ? L.c1 : 0.0;
            S.stage[2 * WAVE + lane] = act ? L.c2 : 0.0;
        });
        ex.each([&](Lane<N> &L, int) { if (L.role != IDLE) { double *sl = slot_ptr<N>(S, SL, L.slot); stage_sum<N>(L, S.stage, sl + SL.SIN, sl + SL.SOUT, D::R2_BASE, D::R2_N, D::R2_BASE, 1); } });
        ex.each([&](Lane<N> &L, int) { border_sigma<N>(L, S, SL); });
        // ---- corrector right-hand side and solve ----
        ex.each([&](Lane<N> &L, int lane) {
            double v3[D::R3_N];
            corrector_rhs<N>(L, S, SL, v3);
            stage_write<D::R3_N, 0, D::R3_N>(v3, S.stage, lane, L.role == BLOCK && !L.done);
        });
        ex.each([&](Lane<N> &L, int) { if (L.role != IDLE) { double *sl = slot_ptr<N>(S, SL, L.slot); stage_sum<N>(L, S.stage, sl + SL.SIN, sl + SL.SOUT, D::R3_BASE, D::R3_N, -1, 0); } });
        ex.each([&](Lane<N> &L, int) { border_corrector_solve<N>(L, S, SL); });
        // ---- final direction: step bound ----
        ex.each([&](Lane<N> &L, int lane) {
            pass_DE<N, false>(L, S, SL, lane, par);
            S.stage[lane] = (L.role == BLOCK && !L.done) ? L.amax : 1e300;
        });
        ex.each([&](Lane<N> &L, int) { if (L.role != IDLE) { double *sl = slot_ptr<N>(S, SL, L.slot); stage_sum<N>(L, S.stage, sl + SL.SIN, sl + SL.SOUT, D::R4_BASE, D::R4_N, D::R4_BASE, 1); } });
        ex.each([&](Lane<N> &L, int) { border_alpha<N>(L, S, SL); });
        // ---- update ----
        ex.each([&](Lane<N> &L, int lane) { pass_DE<N, true>(L, S, SL, lane, par); });
    }
    // ---- un-centre and write out ----
    ex.each([&](Lane<N> &L, int) {
        if (L.role == IDLE) return;
        const double *sl = slot_ptr<N>(S, SL, L.slot);
        const double *cen = sl + SL.CEN;
        if (L.role == BLOCK) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double o1 = L.p[k] + L.yy * cen[k], o2 = L.p[N + k] + L.yy * cen[k];
                a.copy[(size_t)k * a.NI + L.inc] = (T)(L.out ? o1 : L.Tfree[k]);
                a.copy[(size_t)(N + k) * a.NI + L.inc] = (T)(L.out ? o2 : o1);
            }
            a.copy[(size_t)(2 * N) * a.NI + L.inc] = (T)L.yy;
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                a.xv[(size_t)L.v * D::N2 + k] = L.x[k] + cen[k];
                a.xv[(size_t)L.v * D::N2 + N + k] = L.x[N + k] + cen[k];
                a.zv[(size_t)L.v * D::N2 + k] = L.p[k] + L.yy * cen[k];
                a.zv[(size_t)L.v * D::N2 + N + k] = L.p[N + k] + L.yy * cen[k];
            }
            a.yv[L.v] = L.yy;
            ex.count(a.counters, L.status != 0 ? 1 : 0, L.iters);
        }
    });
}

} // namespace gcs
