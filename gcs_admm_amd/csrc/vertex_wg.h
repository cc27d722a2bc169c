// vertex_wg.h -- the WORKGROUP-COOPERATIVE program of the x-update (vertex step): one 256-thread workgroup solves
// one vertex sub-problem (reference: admm_solver_v3.py:352-466 built it, :490 SolveInParallel/MOSEK solved it).
//
// Same interior-point method as the wavefront program (vertex_program.inc) and the oracle -- Mehrotra predictor-
// corrector, Nesterov-Todd scaling of the one cone, arrow elimination blocks -> sides -> reduced border system,
// the numerical rules of DESIGN.md section 3 -- but a different mapping onto the machine:
//   * every matrix and vector of the sub-problem lives in LDS (layout WLay); nothing is held per lane;
//   * each step of the algorithm is a PARALLEL REGION: a loop over independent tasks (one facet row, one matrix
//     entry, one column of an inverse, ...) strided over the 256 threads, closed by a workgroup barrier;
//   * the only serial chains left are the pivots of the Cholesky factorisations (one barrier per column) and the
//     handful of cone scalars thread 0 computes.
// A vertex with d incident edges and m facets has (d+1) "units" (unit 0: the border rows on (z_v, y_v); unit e:
// the block (O_e, y_e)) of 4m facet rows each, so the facet-row passes run (d+1)*4m tasks wide instead of 2m
// iterations deep; the dimension-generic dense algebra (n = 2, 3, 6: blocks of 2n+1, border 4n+1) never touches a
// register array larger than one column.  This is the latency-optimal mapping: it is what small graphs
// (benchmark4: 42 vertices on a 256-CU chip), n = 3 / 6 and vertices of degree > 63 use; large n = 2 graphs keep
// the wavefront program, whose throughput per CU is higher.
//
// The file is host-compilable: with WG_FOR a plain loop and WG_SYNC a no-op the regions execute serially, which is
// exactly equivalent as long as the tasks of a region are independent (tests/hostemu/wg_emu.cpp runs them in
// ascending and in descending order and compares).
#pragma once
#include "gcs_math.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define WG_DEVICE 1
#else
#define WG_DEVICE 0
#endif

namespace gcs_wg {

using gcs_math::rcp;
using gcs_math::rcp1;
using gcs_math::rsqrt_nr;
using gcs_math::sqrt_nr;

constexpr int WG_THREADS = 256;
constexpr double CHOL_SKIP = 1e-12;
constexpr double REG_DELTA = 1e-7;   // Tikhonov term on every centred unknown except t (oracle/gcs_oracle.c REG_DELTA)

#if WG_DEVICE
// The thread id is made opaque at the head of every region: a thread's first task of a region (and its decode into
// unit / row / column, base pointers, ...) is invariant across the Newton loop, and hoisting all of that out of the loop
// costs hundreds of live registers (scratch spills) for no gain.
__device__ __forceinline__ int wg_tid()
{
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
#define WG_FOR(i, cnt) for (int i = gcs_wg::wg_tid(), i##_end = (cnt); i < i##_end; i += gcs_wg::WG_THREADS)
#define WG_SYNC() __syncthreads()
#define WG_ONE() if (gcs_wg::wg_tid() == 0)
#define WG_FENCE() asm volatile("" ::: "memory")
#else
#define WG_FENCE() do { } while (0)
// host: one "thread" runs every task; GCS_WG_REVERSE flips the task order (independence check)
#ifdef GCS_WG_REVERSE
#define WG_FOR(i, cnt) for (int i = (cnt) - 1; i >= 0; --i)
#else
#define WG_FOR(i, cnt) for (int i = 0, i##_end = (cnt); i < i##_end; ++i)
#endif
#define WG_SYNC() do { } while (0)
#define WG_ONE() if (true)
#endif

// diagnostic build (-DGCS_WG_TIMING, tools/wg_phase_timing.py): thread 0 of workgroup 0 accumulates the s_memtime ticks
// between consecutive stamps into g_wg_cycles[id]; nothing of this exists in the product build
#if defined(GCS_WG_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_wg_cycles[64];
__device__ unsigned long long g_wg_counts[64];
#endif
#if defined(GCS_WG_TIMING) && WG_DEVICE
__device__ __forceinline__ void wg_stamp(int id, unsigned long long &last)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_wg_cycles[id], t - last);
        atomicAdd(&g_wg_counts[id], 1ull);
        last = __builtin_amdgcn_s_memtime();
    }
}
#define WG_STAMP(id) gcs_wg::wg_stamp(id, wg_last_stamp)
#define WG_STAMP_INIT() unsigned long long wg_last_stamp = __builtin_amdgcn_s_memtime()
#else
#define WG_STAMP(id) do { } while (0)
#define WG_STAMP_INIT() do { } while (0)
#endif

template <int N> struct WD {
    static constexpr int NW = 2 * N + 1, NX = 2 * N, NB1 = 4 * N + 1, Q = N + 1, NS = N * (N + 1) / 2;
    static constexpr int TA = 2 * NS + 2 * N + 1;   // Hessian assembly tasks per unit
};

// LDS layout of one vertex sub-problem, offsets in doubles.  Everything sits at a COMPILE-TIME offset except the
// stride between units (it holds the 16 m facet-row values of a unit) and the polytope at the end: with ~50 run-time
// offsets live across the Newton loop the compiler ran out of scalar registers and spilled.
//   [ fixed block | unit 0 | unit 1 | ... | unit d | A (m x n) | bc (m) ]
constexpr int pad2(int x) { return (x + 1) & ~1; }      // keep every array 16-byte aligned
template <int N> struct WSoc {   // cone block
    static constexpr int Q = N + 1;
    static constexpr int SS = 0, LS = Q, WB = 2 * Q, LT = 3 * Q, CV = LT + Q, SU = CV + N, DSSA = SU + N * N, DLSA = DSSA + Q,
                         DSS = DLSA + Q, DLS = DSS + Q, KS = DLS + Q, SIZE = KS + Q;
};
enum { SC_T = 0, SC_DT, SC_DTA, SC_ALPHA, SC_CONEFAIL, SC_C0, SC_ETA, SC_GT, SC_N = 8 };
template <int N> struct WL {
    using D = WD<N>;
    static constexpr int NW = D::NW, NX = D::NX, NB1 = D::NB1;
    // fixed block
    static constexpr int CEN = 0, XV = CEN + pad2(N), DX = XV + pad2(NX), NU = DX + pad2(NX), DNU = NU + pad2(2 * NW),
                         GBX = DNU + pad2(2 * NW), BS = GBX + pad2(NX), BSI = BS + pad2(2 * NW * NW), BXS = BSI + pad2(2 * NW * NW),
                         PIVS = BXS + pad2(2 * NW * NX), BG = PIVS + pad2(2 * NW), XBG = BG + pad2(2 * NW), XBX = XBG + pad2(NX),
                         RP = XBX + pad2(NX * NX), VV = RP + pad2(2 * NW), WW = VV + pad2(2 * NW), M = WW + pad2(2 * NW),
                         MINV = M + pad2(NB1 * NB1), PIVM = MINV + pad2(NB1 * NB1), RHS = PIVM + pad2(NB1), SOL = RHS + pad2(NB1),
                         SOC = SOL + pad2(NB1), SC = SOC + pad2(WSoc<N>::SIZE), RED = SC + pad2(SC_N), FIXED = RED + 36;
    // per-unit block (offsets from the unit's base); the four facet-row arrays (4m each) follow at ROWS
    static constexpr int P = 0, DW = P + pad2(NW), TG = DW + pad2(NW), TF = TG + pad2(NW), LB = TF + pad2(N), KB = LB + 2, DLB = KB + 2,
                         PIV = DLB + 2, G0 = PIV + pad2(NW), G = G0 + pad2(NW), GU = G + pad2(NW), GX = GU + pad2(NW), TE = GX + pad2(NX),
                         RV = TE + pad2(NW), K = RV + pad2(NW), X = K + pad2(NW * NW), B = X + pad2(NW * NX), ROWS = B + pad2(NW * NW);
    static GCS_HD int unit_stride(int m) { return ROWS + 16 * m; }
    static GCS_HD int total(int U, int m) { return FIXED + U * unit_stride(m) + pad2(m * N) + pad2(m); }
};
template <int N> GCS_HD int wg_lds_doubles(int U, int m) { return WL<N>::total(U, m); }
inline int wg_lds_doubles_n(int n, int U, int m)
{
    return n == 2 ? wg_lds_doubles<2>(U, m) : (n == 3 ? wg_lds_doubles<3>(U, m) : wg_lds_doubles<6>(U, m));
}

template <class T> struct WgArgs {
    int n_vtx;                  // generic vertices handled by this launch, one workgroup each
    const int *vtx;             // [n_vtx] vertex ids, heaviest first
    const int *inc_ptr;         // [V+1]
    const int *deg_in;          // [V]
    const int *inc_edge;        // [NI_owned]
    const int *poly_ptr;        // [V+1]
    const double *poly_A;       // [sum m][n]
    const double *poly_bc;      // [sum m] centred: b - A c
    const double *center;       // [V][n]
    int E, NI;
    const T *zedge, *mu;
    T *copy;
    double *xv, *zv, *yv;
    int *counters;              // [0] inner failures, [1] inner iterations
    double eps_edge, ipm_tol;
    int ipm_max_iter;
};

// ---------------------------------------------------------------------------------------------------------------
// workgroup reductions: (min, sum, sum) of one value triple per thread -> the same result in every thread
// ---------------------------------------------------------------------------------------------------------------
struct Red3 { double mn, s1, s2; };

#if WG_DEVICE
template <int SH> __device__ __forceinline__ double dpp_row_shr(double x, double ident)
{
    // lane i reads lane i - SH of its 16-lane row; lanes without a source keep `ident`
    const int ilo = __double2loint(ident), ihi = __double2hiint(ident);
    const int lo = __builtin_amdgcn_update_dpp(ilo, __double2loint(x), 0x110 + SH, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(ihi, __double2hiint(x), 0x110 + SH, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double x, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
#endif

GCS_HD Red3 wg_reduce(Red3 v, double *red, int &phase)
{
#if WG_DEVICE
    // rows of 16 lanes with DPP shifts (lane 15 of a row ends up with the row's result), then the four row results
    // through SGPRs, then the four wavefronts through LDS; `red` rotates over three buffers so that one barrier suffices
    v.mn = fmin(v.mn, dpp_row_shr<1>(v.mn, 1e300)); v.s1 += dpp_row_shr<1>(v.s1, 0.0); v.s2 += dpp_row_shr<1>(v.s2, 0.0);
    v.mn = fmin(v.mn, dpp_row_shr<2>(v.mn, 1e300)); v.s1 += dpp_row_shr<2>(v.s1, 0.0); v.s2 += dpp_row_shr<2>(v.s2, 0.0);
    v.mn = fmin(v.mn, dpp_row_shr<4>(v.mn, 1e300)); v.s1 += dpp_row_shr<4>(v.s1, 0.0); v.s2 += dpp_row_shr<4>(v.s2, 0.0);
    v.mn = fmin(v.mn, dpp_row_shr<8>(v.mn, 1e300)); v.s1 += dpp_row_shr<8>(v.s1, 0.0); v.s2 += dpp_row_shr<8>(v.s2, 0.0);
    Red3 w;
    w.mn = fmin(fmin(lane_bcast(v.mn, 15), lane_bcast(v.mn, 31)), fmin(lane_bcast(v.mn, 47), lane_bcast(v.mn, 63)));
    w.s1 = (lane_bcast(v.s1, 15) + lane_bcast(v.s1, 31)) + (lane_bcast(v.s1, 47) + lane_bcast(v.s1, 63));
    w.s2 = (lane_bcast(v.s2, 15) + lane_bcast(v.s2, 31)) + (lane_bcast(v.s2, 47) + lane_bcast(v.s2, 63));
    double *buf = red + phase * 12;
    phase = phase == 2 ? 0 : phase + 1;
    const int wave = (int)threadIdx.x >> 6;
    if (((int)threadIdx.x & 63) == 0) { buf[wave * 3 + 0] = w.mn; buf[wave * 3 + 1] = w.s1; buf[wave * 3 + 2] = w.s2; }
    __syncthreads();
    Red3 r;
    r.mn = fmin(fmin(buf[0], buf[3]), fmin(buf[6], buf[9]));
    r.s1 = (buf[1] + buf[4]) + (buf[7] + buf[10]);
    r.s2 = (buf[2] + buf[5]) + (buf[8] + buf[11]);
    return r;
#else
    (void)red; (void)phase;
    return v;
#endif
}

// small-integer division by a run-time divisor without the ~40-instruction integer sequence (exact for the
// ranges used here: dividend < 2^20, divisor <= 1024; the margin (0.5/div) dwarfs the float rounding)
GCS_HD int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// ---------------------------------------------------------------------------------------------------------------
// cooperative Cholesky: `count` matrices of dimension `dim` (row-major, ld = dim, stride `mstride` between
// matrices), lower triangle in place (strictly lower part = L, diagonal untouched), inverse pivots in piv.
// Pivot rule of oracle chol(): a pivot that has cancelled below CHOL_SKIP of its diagonal entry is clamped there.
// One parallel region per column; task = (matrix, row).
// ---------------------------------------------------------------------------------------------------------------
template <int DIM> GCS_HD void wg_chol(double *mats, double *piv, int count, int mstride, int pstride)
{
    for (int j = 0; j < DIM; ++j) {
        WG_FOR(t, count * DIM) {
            const int q = t / DIM, i = t - q * DIM;
            if (i < j) continue;
            double *Mq = mats + (size_t)q * mstride;
            const double diag = Mq[j * DIM + j];
            double dj = diag, s = Mq[i * DIM + j];
            // all loads of the two row prefixes are issued unconditionally (one LDS round trip per step instead of one per
            // term); terms beyond the prefix are masked out of the arithmetic
#pragma unroll
            for (int k = 0; k < DIM - 1; ++k) {
                const double lj = Mq[j * DIM + k], li = Mq[i * DIM + k];
                const bool use = k < j;
                dj -= use ? lj * lj : 0.0;
                s -= use ? li * lj : 0.0;
            }
            if (!(dj > CHOL_SKIP * diag)) dj = diag > 0.0 ? CHOL_SKIP * diag : 1.0;
            const double inv = rsqrt_nr(dj);
            if (i == j) piv[q * pstride + j] = inv;
            else Mq[i * DIM + j] = s * inv;
        }
        WG_SYNC();
    }
}

// column c of the inverse of the SPD matrix whose factor is (Lm strictly lower, piv inverse pivots): out[i*ldo + c]
// (WG_FENCE: a compiler-level fence per row on the device; without it the scheduler hoists every LDS load of the
//  factor to the top of the unrolled solve and the live ranges spill)
template <int DIM> GCS_HD void chol_inverse_col(const double *Lm, const double *piv, int c, double *out, int ldo)
{
    double x[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= Lm[i * DIM + k] * x[k];
        x[i] = s * piv[i];
        WG_FENCE();
    }
#pragma unroll
    for (int i = DIM - 1; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < DIM; ++k) s -= Lm[k * DIM + i] * x[k];
        x[i] = s * piv[i];
        WG_FENCE();
    }
#pragma unroll
    for (int i = 0; i < DIM; ++i) out[i * ldo + c] = x[i];
}

// the same with the column itself (in LDS) as the work vector, for dimensions whose register copy would spill: per row all
// loads are issued unconditionally (one LDS round trip per row) and the terms outside the triangle are masked out
template <int DIM> GCS_HD void chol_inverse_col_lds(const double *Lm, const double *piv, int c, double *out, int ldo)
{
    for (int i = 0; i < c; ++i) out[i * ldo + c] = 0.0;
    for (int i = c; i < DIM; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < DIM - 1; ++k) {
            const double l = Lm[i * DIM + k], x = out[k * ldo + c];
            s -= (k >= c && k < i) ? l * x : 0.0;
        }
        out[i * ldo + c] = s * piv[i];
        WG_FENCE();
    }
    for (int i = DIM - 1; i >= 0; --i) {
        double s = out[i * ldo + c];
#pragma unroll
        for (int k = 1; k < DIM; ++k) {
            const double l = Lm[k * DIM + i], x = out[k * ldo + c];
            s -= k > i ? l * x : 0.0;
        }
        out[i * ldo + c] = s * piv[i];
        WG_FENCE();
    }
}

// Nesterov-Todd scaling of the cone from (s, z): wb (unit hyperbolic vector), eta; false on a boundary point
template <int Q> GCS_HD bool soc_scaling_wb(const double *s, const double *z, double *wb, double &eta)
{
    const double ss = gcs_math::soc_det<Q>(s), zz = gcs_math::soc_det<Q>(z);
    if (!(ss > 0.0) || !(zz > 0.0)) return false;
    const double is = rsqrt_nr(ss), iz = rsqrt_nr(zz);
    double dot = 0;
    for (int k = 0; k < Q; ++k) dot += (s[k] * is) * (z[k] * iz);
    const double ig2 = 0.5 * rsqrt_nr(0.5 * (1.0 + dot));
    wb[0] = (s[0] * is + z[0] * iz) * ig2;
    for (int k = 1; k < Q; ++k) wb[k] = (s[k] * is - z[k] * iz) * ig2;
    eta = sqrt_nr((ss * is) * iz);
    return true;
}

// products with the Nesterov-Todd scaling W = eta * [wb0 wb1'; wb1 I + wb1 wb1'/(1+wb0)], its inverse and W^{-2} =
// eta^{-2}(2 v v' - J), v = (wb0, -wb1), applied from wb in O(Q) (the explicit Q x Q matrices are never formed)
template <int Q> GCS_HD void soc_apply_W(const double *wb, double eta, const double *x, double *y)
{
    double dd = 0;
#pragma unroll
    for (int k = 1; k < Q; ++k) dd += wb[k] * x[k];
    const double f = x[0] + dd * rcp(1.0 + wb[0]);
    y[0] = eta * (wb[0] * x[0] + dd);
#pragma unroll
    for (int k = 1; k < Q; ++k) y[k] = eta * (x[k] + wb[k] * f);
}
template <int Q> GCS_HD void soc_apply_Wi(const double *wb, double eta, const double *x, double *y)
{
    double dd = 0;
#pragma unroll
    for (int k = 1; k < Q; ++k) dd += wb[k] * x[k];
    const double ieta = rcp(eta), f = -x[0] + dd * rcp(1.0 + wb[0]);
    y[0] = ieta * (wb[0] * x[0] - dd);
#pragma unroll
    for (int k = 1; k < Q; ++k) y[k] = ieta * (x[k] + wb[k] * f);
}
template <int Q> GCS_HD void soc_apply_W2(const double *wb, double eta, const double *x, double *y)
{
    double vx = wb[0] * x[0];
#pragma unroll
    for (int k = 1; k < Q; ++k) vx -= wb[k] * x[k];
    const double ieta = rcp(eta), ie2 = ieta * ieta;
    y[0] = ie2 * (2.0 * wb[0] * vx - x[0]);
#pragma unroll
    for (int k = 1; k < Q; ++k) y[k] = ie2 * (x[k] - 2.0 * wb[k] * vx);
}

// ---------------------------------------------------------------------------------------------------------------
// one vertex sub-problem.  `sm` is the workgroup's LDS (wg_lds_doubles doubles).  Returns (to every thread) the
// solver status (0 = converged) and the number of interior-point iterations through status_out / iters_out.
// ---------------------------------------------------------------------------------------------------------------
template <int N, class T>
GCS_HD void wg_solve_vertex(const WgArgs<T> &a, int v, double rho, double mu_scale, double *sm, int &status_out, int &iters_out)
{
    using D = WD<N>;
    using SO = WSoc<N>;
    using W = WL<N>;
    constexpr int NW = D::NW, NX = D::NX, NB1 = D::NB1, Q = D::Q, NS = D::NS, TA = D::TA;
    const int lo = a.inc_ptr[v], d = a.inc_ptr[v + 1] - lo, d_in = a.deg_in[v], d_out = d - d_in;
    const int p0 = a.poly_ptr[v], m = a.poly_ptr[v + 1] - p0;
    const int U = d + 1, R = 4 * m, RT = U * R, m2 = 2 * m;
    const int US = W::unit_stride(m);
    const float inv_R = 1.0f / (float)R;
    auto UN = [&](int u) -> double * { return sm + W::FIXED + u * US; };       // base of unit u
    const int oS = W::ROWS, oLAM = W::ROWS + R, oR1 = W::ROWS + 2 * R, oR2 = W::ROWS + 3 * R;   // facet-row arrays of a unit
    double *const PA = sm + W::FIXED + U * US;
    const double *A = PA, *BC = PA + pad2(m * N), *CEN = sm + W::CEN;
    double *SOC = sm + W::SOC, *SC = sm + W::SC;
    int red_phase = 0;
    WG_STAMP_INIT();
    const int deg = (4 * m + 2) * (d + 1) + 1;
    const double inv_deg = 1.0 / (double)deg;
    auto side_of = [&](int u) { return (u - 1) >= d_in ? 1 : 0; };   // blocks 1..d_in incoming, the rest outgoing
    auto side_lo = [&](int s) { return s ? d_in + 1 : 1; };
    auto side_hi = [&](int s) { return s ? d : d_in; };              // inclusive

    // ---- load: polytope, targets, start point (strictly feasible, as oracle_solve_vertex) ----
    WG_FOR(t, m * N) PA[t] = a.poly_A[(size_t)p0 * N + t];
    WG_FOR(j, m) PA[pad2(m * N) + j] = a.poly_bc[p0 + j];
    WG_FOR(k, N) sm[W::CEN + k] = a.center[(size_t)v * N + k];
    WG_FOR(t, d * NW) {
        const int e = t / NW, w = t - e * NW, inc = lo + e, edge = a.inc_edge[inc];
        double *un = UN(e + 1);
        const double Tw = (double)a.zedge[(size_t)w * a.E + edge] - mu_scale * (double)a.mu[(size_t)w * a.NI + inc];
        const bool out = e >= d_in;
        // block targets: T1 (of O[:n]), T2 (of O[n:], outgoing only), Ty; the first word of an incoming edge is free
        if (w == 2 * N) un[W::TG + 2 * N] = Tw;
        else if (out) un[W::TG + w] = Tw;
        else if (w < N) { un[W::TF + w] = Tw; un[W::TG + N + w] = 0.0; }
        else un[W::TG + (w - N)] = Tw;
    }
    WG_FOR(t, U * NW) {
        const int u = t / NW, k = t - u * NW;
        double val = 0.0;
        if (k == 2 * N) val = u == 0 ? 0.5 : 0.5 / (double)(side_of(u) ? d_out : d_in);
        UN(u)[W::P + k] = val;
    }
    WG_FOR(k, NX) sm[W::XV + k] = 0.0;
    WG_FOR(k, 2 * NW) sm[W::NU + k] = 0.0;
    WG_ONE() {
        SC[SC_T] = 1.0;
        SOC[SO::LS] = 1.0;
        for (int k = 1; k < Q; ++k) SOC[SO::LS + k] = 0.0;
    }
    WG_SYNC();

    WG_STAMP(0);
    // facet row r of the sub-problem -> (unit, offset inside the unit's row arrays, type a/b, half, facet)
    auto row_decode = [&](int r, int &u, int &ro, int &ty, int &i, int &j) {
        u = fdiv(r, inv_R);
        ro = r - u * R;
        int rem = ro;
        ty = rem >= m2; rem -= ty * m2;
        i = rem >= m; j = rem - i * m;
    };
    // slack direction of a row for the direction (DW of its unit, DX): ds_a = b dy - a.dp_i ; ds_b = -b dy - a.(dx_i - dp_i)
    auto row_ds = [&](const double *un, int ty, int i, int j) {
        double adp = 0, adx = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) { adp += A[j * N + k] * un[W::DW + i * N + k]; adx += A[j * N + k] * sm[W::DX + i * N + k]; }
        const double bdy = BC[j] * un[W::DW + 2 * N];
        return ty == 0 ? bdy - adp : -bdy - (adx - adp);
    };

    // ---- Newton solve with the stored factors for the gradient (G of every unit, GBX, gt): -> DX, DW, DNU, dt (oracle newton_solve) ----
    auto newton_solve = [&](int dt_slot) {
        WG_FOR(t, d * NW) {      // t_e = B_e (-g_e)
            const int u = 1 + t / NW, i = t - (u - 1) * NW;
            double *un = UN(u);
            double s = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) s -= un[W::B + i * NW + k] * un[W::G + k];
            un[W::TE + i] = s;
        }
        WG_SYNC();
        WG_STAMP(30);
        WG_FOR(t, 2 * NW + NX) {   // side sums of t_e; sum of X_e' t_e
            if (t < 2 * NW) {
                const int s = t / NW, i = t - s * NW;
                double acc = 0;
#pragma unroll 4
                for (int u = side_lo(s); u <= side_hi(s); ++u) acc += UN(u)[W::TE + i];
                sm[W::BG + t] = acc;
            } else {
                const int c = t - 2 * NW, h = c / N;
                double acc = 0;
#pragma unroll 2
                for (int u = 1; u <= d; ++u) {
                    const double *un = UN(u);
#pragma unroll
                    for (int k = 0; k < N; ++k) acc += un[W::X + (h * N + k) * NX + c] * un[W::TE + h * N + k];
                    acc += un[W::X + 2 * N * NX + c] * un[W::TE + 2 * N];
                }
                sm[W::XBG + c] = acc;
            }
        }
        WG_SYNC();
        WG_STAMP(31);
        WG_FOR(t, 2 * NW) {      // v_s = Bs^{-1} (rp_s - Bg_s)
            const int s = t / NW, i = t - s * NW;
            const double *Bsi = sm + W::BSI + s * NW * NW;
            double acc = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) acc += Bsi[i * NW + k] * (sm[W::RP + s * NW + k] - sm[W::BG + s * NW + k]);
            sm[W::VV + t] = acc;
        }
        WG_SYNC();
        WG_STAMP(32);
        WG_FOR(q, NB1) {         // right-hand side in the (x, u = z1 - z2, z2, y_v) variables, t eliminated
            const double *g0 = UN(0) + W::G;
            auto base_z = [&](int i) { return -g0[i] - sm[W::VV + i] - sm[W::VV + NW + i]; };
            double r;
            if (q < NX) {
                r = -sm[W::GBX + q] - sm[W::XBG + q];
                for (int s = 0; s < 2; ++s) {
                    const double *BXs = sm + W::BXS + s * NW * NX;
#pragma unroll
                    for (int i = 0; i < NW; ++i) r -= BXs[i * NX + q] * sm[W::VV + s * NW + i];
                }
            } else if (q < NX + N) {          // u rows: r_z1 + cv gt / c0
                const int k = q - NX;
                r = base_z(k) + SOC[SO::CV + k] * SC[SC_GT] * rcp(SC[SC_C0]);
            } else if (q < NX + 2 * N) {      // z2 rows: r_z1 + r_z2 (the cone terms cancel)
                const int k = q - NX - N;
                r = base_z(k) + base_z(N + k);
            } else r = base_z(2 * N);
            sm[W::RHS + q] = r;
        }
        WG_SYNC();
        WG_STAMP(33);
        WG_FOR(q, NB1) {
            const double *Mi = sm + W::MINV + q * NB1;
            double acc = 0;
#pragma unroll
            for (int p = 0; p < NB1; ++p) acc += Mi[p] * sm[W::RHS + p];
            sm[W::SOL + q] = acc;
        }
        WG_SYNC();
        WG_STAMP(34);
        WG_FOR(t, NX + NW + 1) {  // back to (x, z1, z2, y_v), t recovered
            double *u0 = UN(0);
            if (t < NX) sm[W::DX + t] = sm[W::SOL + t];
            else if (t < NX + N) u0[W::DW + (t - NX)] = sm[W::SOL + t] + sm[W::SOL + t + N];     // dz1 = du + dz2
            else if (t < NX + NW) u0[W::DW + (t - NX)] = sm[W::SOL + t];
            else {
                double acc = -SC[SC_GT];
#pragma unroll
                for (int k = 0; k < N; ++k) acc -= SOC[SO::CV + k] * sm[W::SOL + NX + k];
                SC[dt_slot] = acc * rcp(SC[SC_C0]);
            }
        }
        WG_SYNC();
        WG_STAMP(35);
        WG_FOR(t, 2 * NW) {      // w_s = d zeta + (rp_s - Bg_s) + BXs dx
            const int s = t / NW, i = t - s * NW;
            const double *BXs = sm + W::BXS + s * NW * NX;
            double acc = UN(0)[W::DW + i] + (sm[W::RP + t] - sm[W::BG + t]);
#pragma unroll
            for (int c = 0; c < NX; ++c) acc += BXs[i * NX + c] * sm[W::DX + c];
            sm[W::WW + t] = acc;
        }
        WG_SYNC();
        WG_STAMP(36);
        WG_FOR(t, 2 * NW) {      // d nu_s = Bs^{-1} w_s
            const int s = t / NW, i = t - s * NW;
            const double *Bsi = sm + W::BSI + s * NW * NW;
            double acc = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) acc += Bsi[i * NW + k] * sm[W::WW + s * NW + k];
            sm[W::DNU + t] = acc;
        }
        WG_SYNC();
        WG_STAMP(37);
        WG_FOR(t, d * NW) {      // r_e = -g_e + d nu_side - X_e dx
            const int u = 1 + t / NW, i = t - (u - 1) * NW;
            double *un = UN(u);
            double acc = -un[W::G + i] + sm[W::DNU + side_of(u) * NW + i];
            if (i < 2 * N) {
                const int h = i / N;
#pragma unroll
                for (int k = 0; k < N; ++k) acc -= un[W::X + i * NX + h * N + k] * sm[W::DX + h * N + k];
            } else {
#pragma unroll
                for (int c = 0; c < NX; ++c) acc -= un[W::X + i * NX + c] * sm[W::DX + c];
            }
            un[W::RV + i] = acc;
        }
        WG_SYNC();
        WG_STAMP(38);
        WG_FOR(t, d * NW) {      // d w_e = B_e r_e
            const int u = 1 + t / NW, i = t - (u - 1) * NW;
            double *un = UN(u);
            double acc = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) acc += un[W::B + i * NW + k] * un[W::RV + k];
            un[W::DW + i] = acc;
        }
        WG_SYNC();
        WG_STAMP(39);
    };

    int status = -1, it = 0, stalled = 0;
    for (it = 0;; ++it) {
        // ================= rows: slacks, duals at the start, D = l/s, complementarity =================
        double acc = 0.0; int bad = 0;
        WG_FOR(r, RT) {
            int u, ro, ty, i, j;
            row_decode(r, u, ro, ty, i, j);
            double *un = UN(u);
            double ap = 0, ax = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) { ap += A[j * N + k] * un[W::P + i * N + k]; ax += A[j * N + k] * sm[W::XV + i * N + k]; }
            const double yy = un[W::P + 2 * N], b = BC[j];
            const double s = ty == 0 ? b * yy - ap : b * (1.0 - yy) - (ax - ap);
            const double is = rcp1(s);
            if (it == 0) un[oLAM + ro] = is;
            const double l = un[oLAM + ro];
            un[oS + ro] = s;
            un[oR1 + ro] = l * is;
            acc += s * l;
            if (!(s > 0.0)) bad = 1;
        }
        WG_FOR(u, U) {      // bounds 0 <= y <= 1 of every unit
            double *un = UN(u);
            const double yy = un[W::P + 2 * N], s5 = yy, s6 = 1.0 - yy;
            if (it == 0) { un[W::LB] = rcp1(s5); un[W::LB + 1] = rcp1(s6); }
            acc += s5 * un[W::LB] + s6 * un[W::LB + 1];
            if (!(s5 > 0.0) || !(s6 > 0.0)) bad = 1;
        }
        WG_ONE() {          // the cone: s = (t, z1 - z2)
            const double *u0 = UN(0);
            SOC[SO::SS] = SC[SC_T];
            for (int k = 0; k < N; ++k) SOC[SO::SS + 1 + k] = u0[W::P + k] - u0[W::P + N + k];
            for (int k = 0; k < Q; ++k) acc += SOC[SO::SS + k] * SOC[SO::LS + k];
            if (!gcs_math::soc_interior<Q>(SOC + SO::SS)) bad = 1;
        }
        const Red3 r0 = wg_reduce(Red3{bad ? -1.0 : 1.0, acc, 0.0}, sm + W::RED, red_phase);
        WG_STAMP(1);
        const double gap = r0.s1;
        const double mu = r0.mn < 0.0 ? 0.0 / 0.0 : gap * inv_deg;
        {   // stop on the barrier parameter alone (oracle/gcs_oracle.c); a vanishing step = precision exhausted
            const bool conv = mu <= a.ipm_tol || (stalled && mu <= 1e3 * a.ipm_tol);
            bool stop = conv;
            status = conv ? 0 : -1;
            if (!(mu > 0.0)) { stop = true; status = -3; }
            if (!stop && it >= a.ipm_max_iter) { stop = true; status = -1; }
            if (stop) break;
        }

        // ================= Hessian pieces of every unit, objective gradient, cone scaling =================
        WG_FOR(t, 1 + U * TA + U * NW) {
            if (t == 0) {
                // cone scaling: W^{-2} = eta^{-2}(2 v v' - J), v = (wb0, -wb1); t is eliminated in closed form (c0, cv, Su):
                // a numerical pivot on t cancels catastrophically once the cone is active
                double wb[Q], eta = 1.0;
                if (!soc_scaling_wb<Q>(SOC + SO::SS, SOC + SO::LS, wb, eta)) { SC[SC_CONEFAIL] = 1.0; continue; }
                SC[SC_CONEFAIL] = 0.0; SC[SC_ETA] = eta;
                const double ieta = rcp(eta), ie2 = ieta * ieta;
#pragma unroll
                for (int i = 0; i < Q; ++i) SOC[SO::WB + i] = wb[i];
                {
                    double ls[Q], lt[Q];
#pragma unroll
                    for (int i = 0; i < Q; ++i) ls[i] = SOC[SO::LS + i];
                    soc_apply_W<Q>(wb, eta, ls, lt);
#pragma unroll
                    for (int i = 0; i < Q; ++i) SOC[SO::LT + i] = lt[i];
                }
                const double den = 2.0 * wb[0] * wb[0] - 1.0, g2 = 2.0 * rcp(den);
                SC[SC_C0] = ie2 * den;
                for (int k = 0; k < N; ++k) {
                    SOC[SO::CV + k] = -ie2 * 2.0 * wb[0] * wb[1 + k];
                    for (int l = 0; l < N; ++l) SOC[SO::SU + k * N + l] = ie2 * ((k == l ? 1.0 : 0.0) - g2 * wb[1 + k] * wb[1 + l]);
                }
                continue;
            }
            const int ta = t - 1;
            if (ta < U * TA) {
                // K_u = [K1 0 k1y; 0 K2 k2y; . . kyy] (rows a+b), X_u = d(unit)/d(x) coupling (rows b); reference rows
                // admm_solver_v3.py:420-426 (unit 0) and :434-440 (blocks)
                const int u = ta / TA, q = ta - u * TA;
                double *un = UN(u);
                double *K = un + W::K, *X = un + W::X;
                const double *Da = un + oR1, *Db = Da + m2;
                const bool blk = u > 0, out = blk && side_of(u);
                if (q < 2 * NS) {
                    const int i = q / NS, pq = q - i * NS;
                    int k = 0;
                    while ((k + 1) * (k + 2) / 2 <= pq) ++k;
                    const int l = pq - k * (k + 1) / 2;
                    double sk = 0, sx = 0;
#pragma unroll 4
                    for (int j = 0; j < m; ++j) {
                        const double aa = A[j * N + k] * A[j * N + l];
                        sk += (Da[i * m + j] + Db[i * m + j]) * aa;
                        sx += Db[i * m + j] * aa;
                    }
                    if (k == l) { sk += REG_DELTA; if (blk && (i == 0 || out)) sk += rho; }
                    K[(i * N + k) * NW + i * N + l] = sk; K[(i * N + l) * NW + i * N + k] = sk;
                    X[(i * N + k) * NX + i * N + l] = -sx; X[(i * N + l) * NX + i * N + k] = -sx;
                    const int o = (1 - i) * N;      // the two halves are not coupled directly
                    K[(i * N + k) * NW + o + l] = 0.0; K[(i * N + l) * NW + o + k] = 0.0;
                    X[(i * N + k) * NX + o + l] = 0.0; X[(i * N + l) * NX + o + k] = 0.0;
                } else if (q < 2 * NS + 2 * N) {
                    const int ik = q - 2 * NS, i = ik / N, k = ik - i * N;
                    double sk = 0, sx = 0;
#pragma unroll 4
                    for (int j = 0; j < m; ++j) {
                        const double ba = BC[j] * A[j * N + k];
                        sk -= (Da[i * m + j] + Db[i * m + j]) * ba;
                        sx += Db[i * m + j] * ba;
                    }
                    if (blk && (i == 0 || out)) sk += rho * CEN[k];
                    K[(i * N + k) * NW + 2 * N] = sk; K[2 * N * NW + i * N + k] = sk;
                    X[2 * N * NX + i * N + k] = sx;
                } else {
                    double sk = 0;
#pragma unroll 4
                    for (int j = 0; j < m2; ++j) { const double b = BC[j >= m ? j - m : j]; sk += (Da[j] + Db[j]) * b * b; }
                    const double yy = un[W::P + 2 * N];
                    sk += un[W::LB] * rcp1(yy) + un[W::LB + 1] * rcp1(1.0 - yy) + REG_DELTA;
                    if (blk) {
                        double cc = 0;
#pragma unroll
                        for (int k = 0; k < N; ++k) cc += CEN[k] * CEN[k];
                        sk += rho * (1.0 + (out ? 2.0 : 1.0) * cc);
                    }
                    K[2 * N * NW + 2 * N] = sk;
                }
            } else {
                // gradient of the smooth objective + equality multipliers + Tikhonov term (no facet-row part):
                // blocks: consensus penalty (admm_solver_v3.py:392-413) and the edge cost 1e-4 y_e (:387-388)
                const int tg = ta - U * TA, u = tg / NW, k = tg - u * NW;
                double *un = UN(u);
                const double *p = un + W::P;
                double g;
                if (u == 0) g = sm[W::NU + k] + sm[W::NU + NW + k] + REG_DELTA * p[k];
                else {
                    const bool out = side_of(u);
                    const double *tg_ = un + W::TG, *nu = sm + W::NU + (out ? NW : 0);
                    const double yy = p[2 * N];
                    if (k < N) g = rho * (p[k] + yy * CEN[k] - tg_[k]);
                    else if (k < 2 * N) g = out ? rho * (p[k] + yy * CEN[k - N] - tg_[k]) : 0.0;
                    else {
                        g = rho * (yy - tg_[2 * N]) + a.eps_edge;
#pragma unroll
                        for (int c = 0; c < N; ++c) {
                            g += CEN[c] * rho * (p[c] + yy * CEN[c] - tg_[c]);
                            if (out) g += CEN[c] * rho * (p[N + c] + yy * CEN[c] - tg_[N + c]);
                        }
                    }
                    g += REG_DELTA * p[k] - nu[k];
                }
                un[W::G0 + k] = g;
                un[W::G + k] = g;
            }
        }
        WG_FOR(c, NX) sm[W::GBX + c] = REG_DELTA * sm[W::XV + c];
        WG_ONE() SC[SC_GT] = 1.0;
        WG_SYNC();
        WG_STAMP(2);
        if (SC[SC_CONEFAIL] != 0.0) { status = mu <= 1e3 * a.ipm_tol ? 0 : -4; break; }

        // ================= blocks: Cholesky, explicit inverse B_e, B_e X_e =================
        wg_chol<NW>(UN(1) + W::K, UN(1) + W::PIV, d, US, US);
        WG_STAMP(3);
        WG_FOR(t, d * NW) {
            const int u = 1 + t / NW, c = t - (u - 1) * NW;
            double *un = UN(u);
            chol_inverse_col<NW>(un + W::K, un + W::PIV, c, un + W::B, NW);
        }
        WG_SYNC();
        WG_STAMP(4);
        WG_FOR(t, d * NW * NX) {     // B_e X_e, written over the (now dead) factor of the block
            const int u = 1 + t / (NW * NX), ic = t - (u - 1) * (NW * NX), i = ic / NX, c = ic - i * NX, h = c / N;
            double *un = UN(u);
            double s = un[W::B + i * NW + 2 * N] * un[W::X + 2 * N * NX + c];
#pragma unroll
            for (int k = 0; k < N; ++k) s += un[W::B + i * NW + h * N + k] * un[W::X + (h * N + k) * NX + c];
            un[W::K + ic] = s;
        }
        WG_SYNC();
        WG_STAMP(5);
        // ================= side sums: Bs, BXs, X'BX, equality residuals =================
        WG_FOR(t, 2 * NW * NW + 2 * NW * NX + NX * NX + 2 * NW) {
            if (t < 2 * NW * NW) {
                const int s = t / (NW * NW), q = t - s * NW * NW;
                double acc2 = 0;
#pragma unroll 4
                for (int u = side_lo(s); u <= side_hi(s); ++u) acc2 += UN(u)[W::B + q];
                sm[W::BS + t] = acc2;
            } else if (t < 2 * NW * NW + 2 * NW * NX) {
                const int tt = t - 2 * NW * NW, s = tt / (NW * NX), q = tt - s * NW * NX;
                double acc2 = 0;
#pragma unroll 4
                for (int u = side_lo(s); u <= side_hi(s); ++u) acc2 += UN(u)[W::K + q];
                sm[W::BXS + tt] = acc2;
            } else if (t < 2 * NW * NW + 2 * NW * NX + NX * NX) {
                const int tt = t - 2 * NW * NW - 2 * NW * NX, r = tt / NX, c = tt - r * NX, h = r / N;
                double acc2 = 0;
#pragma unroll 2
                for (int u = 1; u <= d; ++u) {
                    const double *un = UN(u);
#pragma unroll
                    for (int k = 0; k < N; ++k) acc2 += un[W::X + (h * N + k) * NX + r] * un[W::K + (h * N + k) * NX + c];
                    acc2 += un[W::X + 2 * N * NX + r] * un[W::K + 2 * N * NX + c];
                }
                sm[W::XBX + tt] = acc2;
            } else {
                const int tt = t - 2 * NW * NW - 2 * NW * NX - NX * NX, s = tt / NW, k = tt - s * NW;
                double acc2 = UN(0)[W::P + k];
#pragma unroll 4
                for (int u = side_lo(s); u <= side_hi(s); ++u) acc2 -= UN(u)[W::P + k];
                sm[W::RP + tt] = acc2;
            }
        }
        WG_SYNC();
        WG_STAMP(6);
        // ================= sides: factor, invert, Y_s = Bs^{-1} BXs (over the dead factor) =================
        wg_chol<NW>(sm + W::BS, sm + W::PIVS, 2, NW * NW, NW);
        WG_STAMP(7);
        WG_FOR(t, 2 * NW) {
            const int s = t / NW, c = t - s * NW;
            chol_inverse_col<NW>(sm + W::BS + s * NW * NW, sm + W::PIVS + s * NW, c, sm + W::BSI + s * NW * NW, NW);
        }
        WG_SYNC();
        WG_FOR(t, 2 * NW * NX) {
            const int s = t / (NW * NX), ic = t - s * NW * NX, i = ic / NX, c = ic - i * NX;
            const double *Bsi = sm + W::BSI + s * NW * NW, *BXs = sm + W::BXS + s * NW * NX;
            double acc2 = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) acc2 += Bsi[i * NW + k] * BXs[k * NX + c];
            sm[W::BS + s * NW * NW + ic] = acc2;     // Y_s
        }
        WG_SYNC();
        WG_STAMP(9);
        // ================= reduced border matrix in the (x, u, z2, y_v) variables =================
        {
            const double *YS0 = sm + W::BS, *YS1 = sm + W::BS + NW * NW, *u0 = UN(0);
            // entry (r, c) of the matrix in the (x, z1, z2, y_v) variables
            auto mval = [&](int r, int c) -> double {
                if (r < c) { const int t_ = r; r = c; c = t_; }
                if (r < NX) {                    // x-x
                    double acc2 = (r == c ? REG_DELTA : 0.0) - sm[W::XBX + r * NX + c];
                    if (r / N == c / N)
#pragma unroll 4
                        for (int u = 0; u <= d; ++u) acc2 -= UN(u)[W::X + r * NX + c];
#pragma unroll
                    for (int k = 0; k < NW; ++k)
                        acc2 += sm[W::BXS + k * NX + r] * YS0[k * NX + c] + sm[W::BXS + NW * NX + k * NX + r] * YS1[k * NX + c];
                    return acc2;
                }
                if (c < NX) {                    // zeta-x
                    const int i = r - NX;
                    return u0[W::X + i * NX + c] + YS0[i * NX + c] + YS1[i * NX + c];
                }
                const int i = r - NX, k = c - NX; // zeta-zeta
                return u0[W::K + i * NW + k] + sm[W::BSI + i * NW + k] + sm[W::BSI + NW * NW + i * NW + k];
            };
            WG_FOR(t, NB1 * (NB1 + 1) / 2) {
                int r = 0;
                while ((r + 1) * (r + 2) / 2 <= t) ++r;
                const int c = t - r * (r + 1) / 2;
                // change of variables (u, z2) = (z1 - z2, z2): columns/rows of z2 gain those of z1; the cone term Su then
                // sits on u alone (with the cone inactive Su ~ 1/mu would cancel in the (z1, z2) form)
                const bool rz2 = r >= NX + N && r < NX + 2 * N, cz2 = c >= NX + N && c < NX + 2 * N;
                double val = mval(r, c);
                if (cz2) val += mval(r, c - N);
                if (rz2) val += mval(r - N, c);
                if (rz2 && cz2) val += mval(r - N, c - N);
                if (r >= NX && r < NX + N && c >= NX && c < NX + N) val += SOC[SO::SU + (r - NX) * N + (c - NX)];
                sm[W::M + r * NB1 + c] = val;
            }
        }
        WG_SYNC();
        WG_STAMP(10);
        wg_chol<NB1>(sm + W::M, sm + W::PIVM, 1, NB1 * NB1, NB1);
        WG_STAMP(11);
        WG_FOR(c, NB1) {
            if constexpr (NB1 <= 13) chol_inverse_col<NB1>(sm + W::M, sm + W::PIVM, c, sm + W::MINV, NB1);
            else chol_inverse_col_lds<NB1>(sm + W::M, sm + W::PIVM, c, sm + W::MINV, NB1);
        }
        WG_SYNC();

        WG_STAMP(12);
        // ================= affine direction (kappa = 0) =================
        newton_solve(SC_DTA);
        WG_STAMP(13);
        // rows: step bound, mu_aff sums, ds_a dl_a
        double rmax = 0.0, c1 = 0.0, c2 = 0.0;
        WG_FOR(r, RT) {
            int u, ro, ty, i, j;
            row_decode(r, u, ro, ty, i, j);
            double *un = UN(u);
            const double s = un[oS + ro], l = un[oLAM + ro], is = rcp1(s);
            const double ds = row_ds(un, ty, i, j);
            const double q = ds * is, dl = -l - l * q;          // dl / l = -1 - ds / s
            rmax = fmax(rmax, fmax(-q, 1.0 + q));
            c1 += s * dl + l * ds; c2 += ds * dl;
            un[oR1 + ro] = ds * dl;
        }
        WG_FOR(u, U) {
            double *un = UN(u);
            const double yy = un[W::P + 2 * N], dy = un[W::DW + 2 * N];
            const double s5 = yy, s6 = 1.0 - yy, l5 = un[W::LB], l6 = un[W::LB + 1];
            const double q5 = dy * rcp1(s5), q6 = -dy * rcp1(s6);
            const double dl5 = -l5 - l5 * q5, dl6 = -l6 - l6 * q6;
            rmax = fmax(rmax, fmax(fmax(-q5, 1.0 + q5), fmax(-q6, 1.0 + q6)));
            c1 += s5 * dl5 + l5 * dy + s6 * dl6 - l6 * dy; c2 += dy * dl5 - dy * dl6;
            un[W::KB] = dy * dl5; un[W::KB + 1] = -dy * dl6;
        }
        double amax_cone = 1e300;
        WG_ONE() {
            const double *u0 = UN(0);
            SOC[SO::DSSA] = SC[SC_DTA];
            for (int k = 0; k < N; ++k) SOC[SO::DSSA + 1 + k] = u0[W::DW + k] - u0[W::DW + N + k];
            {
                double wb[Q], xs[Q], ys[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) { wb[k] = SOC[SO::WB + k]; xs[k] = SOC[SO::DSSA + k]; }
                soc_apply_W2<Q>(wb, SC[SC_ETA], xs, ys);
#pragma unroll
                for (int k = 0; k < Q; ++k) SOC[SO::DLSA + k] = -SOC[SO::LS + k] - ys[k];
            }
            amax_cone = fmin(gcs_math::soc_max_step<Q>(SOC + SO::SS, SOC + SO::DSSA), gcs_math::soc_max_step<Q>(SOC + SO::LS, SOC + SO::DLSA));
            for (int k = 0; k < Q; ++k) {
                c1 += SOC[SO::SS + k] * SOC[SO::DLSA + k] + SOC[SO::LS + k] * SOC[SO::DSSA + k];
                c2 += SOC[SO::DSSA + k] * SOC[SO::DLSA + k];
            }
        }
        const Red3 rb = wg_reduce(Red3{fmin(rmax > 0.0 ? rcp1(rmax) : 1e300, amax_cone), c1, c2}, sm + W::RED, red_phase);
        WG_STAMP(14);
        double sigmu;
        {
            const double al = fmin(1.0, rb.mn);
            const double mu_aff = (gap + al * rb.s1 + al * al * rb.s2) * inv_deg;
            double sig = mu_aff * rcp(mu);
            sig = sig < 0 ? 0 : (sig > 1 ? 1 : sig);
            sig = sig * sig * sig;
            sigmu = sig * mu;
        }
        // ================= corrector: kappa = (sigma mu - ds_a dl_a) / s per row, cone part by thread 0 =================
        WG_FOR(r, RT) {
            const int u = fdiv(r, inv_R), ro = r - u * R;
            double *un = UN(u);
            un[oR2 + ro] = (sigmu - un[oR1 + ro]) * rcp1(un[oS + ro]);
        }
        WG_FOR(u, U) {
            double *un = UN(u);
            const double yy = un[W::P + 2 * N];
            un[W::KB] = (sigmu - un[W::KB]) * rcp1(yy);
            un[W::KB + 1] = (sigmu - un[W::KB + 1]) * rcp1(1.0 - yy);
        }
        WG_ONE() {   // kappa_soc = sigma mu s^{-1} - W^{-1}( lt \ ((W^{-1} ds_a) o (W dl_a)) )
            double wb[Q], a1[Q], a2[Q], pr[Q], qv[Q], xs[Q], lt[Q], ss[Q];
            const double eta = SC[SC_ETA];
#pragma unroll
            for (int k = 0; k < Q; ++k) { wb[k] = SOC[SO::WB + k]; xs[k] = SOC[SO::DSSA + k]; lt[k] = SOC[SO::LT + k]; ss[k] = SOC[SO::SS + k]; }
            soc_apply_Wi<Q>(wb, eta, xs, a1);
#pragma unroll
            for (int k = 0; k < Q; ++k) xs[k] = SOC[SO::DLSA + k];
            soc_apply_W<Q>(wb, eta, xs, a2);
            double dsum = 0;
#pragma unroll
            for (int k = 0; k < Q; ++k) dsum += a1[k] * a2[k];
            pr[0] = dsum;
#pragma unroll
            for (int k = 1; k < Q; ++k) pr[k] = a1[0] * a2[k] + a2[0] * a1[k];
            const double det = gcs_math::soc_det<Q>(lt);
            double ld1 = 0;
#pragma unroll
            for (int k = 1; k < Q; ++k) ld1 += lt[k] * pr[k];
            qv[0] = (lt[0] * pr[0] - ld1) * rcp(det);
            const double ilt0 = rcp(lt[0]);
#pragma unroll
            for (int k = 1; k < Q; ++k) qv[k] = (pr[k] - qv[0] * lt[k]) * ilt0;
            const double smd = sigmu * rcp(gcs_math::soc_det<Q>(ss));
            soc_apply_Wi<Q>(wb, eta, qv, a1);
#pragma unroll
            for (int i = 0; i < Q; ++i) SOC[SO::KS + i] = smd * (i == 0 ? ss[0] : -ss[i]) - a1[i];
            SC[SC_GT] = 1.0 - SOC[SO::KS];
        }
        WG_SYNC();
        WG_STAMP(15);
        // G' kappa per unit: own unknowns (GU) and the x part (GX)
        WG_FOR(t, U * (NW + NX)) {
            const int u = t / (NW + NX), q = t - u * (NW + NX);
            double *un = UN(u);
            const double *ka = un + oR2, *kb = ka + m2;
            double s = 0;
            if (q < 2 * N) {
                const int i = q / N, k = q - i * N;
#pragma unroll 4
                for (int j = 0; j < m; ++j) s += A[j * N + k] * (ka[i * m + j] - kb[i * m + j]);
                un[W::GU + q] = s;
            } else if (q == 2 * N) {
#pragma unroll 4
                for (int j = 0; j < m2; ++j) s += BC[j >= m ? j - m : j] * (kb[j] - ka[j]);
                un[W::GU + q] = s - un[W::KB] + un[W::KB + 1];
            } else {
                const int c = q - NW, i = c / N, k = c - i * N;
#pragma unroll 4
                for (int j = 0; j < m; ++j) s += A[j * N + k] * kb[i * m + j];
                un[W::GX + c] = s;
            }
        }
        WG_SYNC();
        WG_STAMP(16);
        WG_FOR(t, U * NW + NX) {
            if (t < U * NW) {
                const int u = t / NW, k = t - u * NW;
                double *un = UN(u);
                double g = un[W::G0 + k] + un[W::GU + k];
                if (u == 0 && k < N) g -= SOC[SO::KS + 1 + k];
                else if (u == 0 && k < 2 * N) g += SOC[SO::KS + 1 + (k - N)];
                un[W::G + k] = g;
            } else {
                const int c = t - U * NW;
                double g = REG_DELTA * sm[W::XV + c];
#pragma unroll 4
                for (int u = 0; u <= d; ++u) g += UN(u)[W::GX + c];
                sm[W::GBX + c] = g;
            }
        }
        WG_SYNC();
        WG_STAMP(17);
        newton_solve(SC_DT);
        WG_STAMP(18);
        // ================= final direction: dual directions, step bound =================
        rmax = 0.0;
        WG_FOR(r, RT) {
            int u, ro, ty, i, j;
            row_decode(r, u, ro, ty, i, j);
            double *un = UN(u);
            const double s = un[oS + ro], l = un[oLAM + ro];
            const double ip = rcp1(s * l), is = l * ip, il = s * ip;      // 1/s and 1/l from one reciprocal
            const double ds = row_ds(un, ty, i, j);
            const double dl = un[oR2 + ro] - l - (l * is) * ds;
            un[oR2 + ro] = dl;
            rmax = fmax(rmax, fmax(-ds * is, -dl * il));
        }
        WG_FOR(u, U) {
            double *un = UN(u);
            const double yy = un[W::P + 2 * N], dy = un[W::DW + 2 * N];
            const double s5 = yy, s6 = 1.0 - yy, l5 = un[W::LB], l6 = un[W::LB + 1];
            const double i5 = rcp1(s5), i6 = rcp1(s6);
            const double dl5 = un[W::KB] - l5 - l5 * i5 * dy, dl6 = un[W::KB + 1] - l6 + l6 * i6 * dy;
            un[W::DLB] = dl5; un[W::DLB + 1] = dl6;
            rmax = fmax(rmax, fmax(fmax(-dy * i5, -dl5 * rcp1(l5)), fmax(dy * i6, -dl6 * rcp1(l6))));
        }
        amax_cone = 1e300;
        WG_ONE() {
            const double *u0 = UN(0);
            SOC[SO::DSS] = SC[SC_DT];
            for (int k = 0; k < N; ++k) SOC[SO::DSS + 1 + k] = u0[W::DW + k] - u0[W::DW + N + k];
            {
                double wb[Q], xs[Q], ys[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) { wb[k] = SOC[SO::WB + k]; xs[k] = SOC[SO::DSS + k]; }
                soc_apply_W2<Q>(wb, SC[SC_ETA], xs, ys);
#pragma unroll
                for (int k = 0; k < Q; ++k) SOC[SO::DLS + k] = SOC[SO::KS + k] - SOC[SO::LS + k] - ys[k];
            }
            amax_cone = fmin(gcs_math::soc_max_step<Q>(SOC + SO::SS, SOC + SO::DSS), gcs_math::soc_max_step<Q>(SOC + SO::LS, SOC + SO::DLS));
        }
        const Red3 rd = wg_reduce(Red3{fmin(rmax > 0.0 ? rcp1(rmax) : 1e300, amax_cone), 0.0, 0.0}, sm + W::RED, red_phase);
        WG_STAMP(19);
        WG_ONE() {      // step length with the cone guard (round-off must not push either cone point outside)
            double al = fmin(1.0, 0.99 * rd.mn);
            for (int tries = 0; tries < 40; ++tries) {
                double s2[Q], l2[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) { s2[k] = SOC[SO::SS + k] + al * SOC[SO::DSS + k]; l2[k] = SOC[SO::LS + k] + al * SOC[SO::DLS + k]; }
                if (gcs_math::soc_interior<Q>(s2) && gcs_math::soc_interior<Q>(l2)) break;
                al *= 0.7;
            }
            SC[SC_ALPHA] = al;
        }
        WG_SYNC();
        WG_STAMP(20);
        const double alpha = SC[SC_ALPHA];
        stalled = alpha < 1e-3;
        // ================= update =================
        WG_FOR(r, RT) {
            const int u = fdiv(r, inv_R), ro = r - u * R;
            double *un = UN(u);
            un[oLAM + ro] += alpha * un[oR2 + ro];
        }
        WG_FOR(t, U * NW) {
            const int u = t / NW, k = t - u * NW;
            double *un = UN(u);
            un[W::P + k] += alpha * un[W::DW + k];
        }
        WG_FOR(t, 2 * U) {
            double *un = UN(t >> 1);
            un[W::LB + (t & 1)] += alpha * un[W::DLB + (t & 1)];
        }
        WG_FOR(c, NX) sm[W::XV + c] += alpha * sm[W::DX + c];
        WG_FOR(t, 2 * NW) sm[W::NU + t] += alpha * sm[W::DNU + t];
        WG_ONE() {
            SC[SC_T] += alpha * SC[SC_DT];
            for (int k = 0; k < Q; ++k) SOC[SO::LS + k] += alpha * SOC[SO::DLS + k];
        }
        WG_SYNC();
        WG_STAMP(21);
    }
    status_out = status;
    iters_out = it;
    WG_STAMP(22);
    WG_SYNC();
    if (status != 0) return;      // inner failure: the previous copy columns stay (admm_solver_v3.py:524-538 intent)
    // ---- un-centre and write out ----
    WG_FOR(t, d * NW) {
        const int e = t / NW, w = t - e * NW, inc = lo + e;
        const bool out = e >= d_in;
        const double *un = UN(e + 1), *p = un + W::P;
        const double yy = p[2 * N];
        double val;
        if (w == 2 * N) val = yy;
        else if (w < N) val = out ? p[w] + yy * CEN[w] : un[W::TF + w];     // incoming: the free word keeps its target
        else val = out ? p[w] + yy * CEN[w - N] : p[w - N] + yy * CEN[w - N];
        a.copy[(size_t)w * a.NI + inc] = (T)val;
    }
    WG_FOR(k, NX) {
        const int c = k < N ? k : k - N;
        const double *u0 = UN(0);
        a.xv[(size_t)v * NX + k] = sm[W::XV + k] + CEN[c];
        a.zv[(size_t)v * NX + k] = u0[W::P + k] + u0[W::P + 2 * N] * CEN[c];
    }
    WG_ONE() a.yv[v] = UN(0)[W::P + 2 * N];
}

} // namespace gcs_wg
