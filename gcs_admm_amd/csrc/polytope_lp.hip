// polytope_lp.hip -- batched tiny linear programs for graph construction at scale (gfx950)
//
// The reference builds the graph with one LP feasibility solve per ordered pair of regions through
// Drake/MOSEK (utils.py:31-82: build_graph -> check_overlap, :49-65) -- |V|^2 host solves.  Here one
// lane solves one LP with a primal-dual interior-point method (Mehrotra predictor-corrector, normal
// equations of size n+1 <= 7, f64), 64 LPs per wavefront, rows streamed from the polytope CSR:
//
//   centres  : max r  s.t.  a_i x + r |a_i| <= b_i                  (Chebyshev centre of one polytope:
//                                                                    the interior point the vertex kernel
//                                                                    centres its sub-problem on)
//   overlaps : the same LP over the rows of two polytopes; the pair intersects iff r* >= -tol
//              (closed sets: touching counts, as it does for an LP feasibility solve)
//   bounds   : min / max x_k over one polytope, from its centre      (axis-aligned bounding boxes for the
//                                                                    broad phase: sort-and-sweep on the host)
//
// Per lane: the unknowns, the (n+1)^2 normal matrix and its Cholesky factor live in registers; the row
// duals and their directions sit in LDS as [row][lane] (conflict-free); every Newton iteration makes five
// passes over the rows.  An overlap LP stops as soon as the current (always strictly feasible) iterate has
// r > 0, or the dual bound proves r* < -tol.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "gcsadmm.h"

#include "polytope_lp_core.h"

namespace gcsadmm_lp {

// ---- kernels (LDS: lam[maxm][64], dlam[maxm][64]) ----
template <int N>
__global__ __launch_bounds__(WAVE) void ball_kernel(Polys S, int count, const int *pa, const int *pb, const double *x0s,
                                                   int maxm, double tol, int early, double *out_w, unsigned char *out_flag, int *out_status)
{
    extern __shared__ double smem[];
    double *lam = smem, *dlam = smem + (size_t)maxm * WAVE;
    const int t = blockIdx.x * WAVE + threadIdx.x, lane = threadIdx.x;
    if (t >= count) return;
    const int p = pa ? pa[t] : t, q = pb ? pb[t] : -1;
    Rows<N, true> R(S, p, q);
    double w[N + 1], c[N + 1];
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = 0.0;
    c[N] = -1.0;
    ball_start<N>(R, x0s ? x0s + (size_t)p * N : nullptr, w);
    int iters = 0;
    const int st = lp_ipm<N, true>(R, c, w, lam, dlam, lane, early != 0, tol, &iters);
    if (out_w) {
#pragma unroll
        for (int k = 0; k <= N; ++k) out_w[(size_t)t * (N + 1) + k] = w[k];
    }
    if (out_flag) out_flag[t] = (st == 1) ? 1 : (st == 2 ? 0 : (w[N] >= -tol ? 1 : 0));
    if (out_status) out_status[t] = st;
}

template <int N>
__global__ __launch_bounds__(WAVE) void bounds_kernel(Polys S, const double *centers, int maxm, double *lo, double *hi, int *out_status)
{
    extern __shared__ double smem[];
    double *lam = smem, *dlam = smem + (size_t)maxm * WAVE;
    const int t = blockIdx.x * WAVE + threadIdx.x, lane = threadIdx.x;
    if (t >= S.P * 2 * N) return;
    const int p = t / (2 * N), j = t % (2 * N), k = j >> 1, upper = j & 1;
    Rows<N, false> R(S, p, -1);
    double w[N], c[N];
#pragma unroll
    for (int kk = 0; kk < N; ++kk) { w[kk] = centers[(size_t)p * N + kk]; c[kk] = 0.0; }
    c[k] = upper ? -1.0 : 1.0;
    const int st = lp_ipm<N, false>(R, c, w, lam, dlam, lane, false, 0.0, nullptr);
    (upper ? hi : lo)[(size_t)p * N + k] = w[k];
    if (out_status) out_status[t] = st;
}

static std::string g_err;

// the entry points switch to the requested device (upload_scene) and hand the caller's current device back on return
struct RestoreDevice {
    int prev = -1;
    RestoreDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~RestoreDevice()
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() { return (T *)p; }
};

struct Scene {
    DevBuf ptr, A, b, nrm;
    Polys S;
    int maxm = 0;
};

static int upload_scene(Scene &sc, int n, int P, const int *poly_ptr, const double *A, const double *b, int device)
{
    if (n < 1 || n > 8) { g_err = "polytope LPs are instantiated for n = 1..8"; return GCSADMM_ERR_UNSUPPORTED; }
    if (P < 0 || !poly_ptr || (P > 0 && (!A || !b))) { g_err = "null polytope array"; return GCSADMM_ERR_BAD_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device"; return GCSADMM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { g_err = "device ordinal out of range"; return GCSADMM_ERR_BAD_ARG; }
    if (poly_ptr[0] != 0) { g_err = "poly_ptr[0] != 0"; return GCSADMM_ERR_BAD_ARG; }
    for (int p = 0; p < P; ++p) {
        const int m = poly_ptr[p + 1] - poly_ptr[p];
        if (m < 1) { g_err = "polytope without rows"; return GCSADMM_ERR_BAD_ARG; }
        sc.maxm = std::max(sc.maxm, m);
    }
    const int rows = poly_ptr[P];
    std::vector<double> nrm((size_t)rows);
    for (int r = 0; r < rows; ++r) {
        double s = 0;
        for (int k = 0; k < n; ++k) s += A[(size_t)r * n + k] * A[(size_t)r * n + k];
        if (!(s > 0.0)) { g_err = "zero facet normal"; return GCSADMM_ERR_BAD_ARG; }
        nrm[r] = std::sqrt(s);
    }
    hipError_t e;
#define CK(x) if ((e = (x)) != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e); return GCSADMM_ERR_HIP; }
    CK(hipSetDevice(device));
    CK(sc.ptr.alloc(sizeof(int) * (P + 1))); CK(sc.A.alloc(sizeof(double) * rows * n));
    CK(sc.b.alloc(sizeof(double) * rows)); CK(sc.nrm.alloc(sizeof(double) * rows));
    CK(hipMemcpy(sc.ptr.p, poly_ptr, sizeof(int) * (P + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(sc.A.p, A, sizeof(double) * rows * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc.b.p, b, sizeof(double) * rows, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc.nrm.p, nrm.data(), sizeof(double) * rows, hipMemcpyHostToDevice));
    sc.S = Polys{n, P, sc.ptr.as<int>(), sc.A.as<double>(), sc.b.as<double>(), sc.nrm.as<double>()};
    return GCSADMM_OK;
}

template <int N>
static int launch_ball(const Scene &sc, long count, const int *d_pa, const int *d_pb, const double *d_x0, int rows_max,
                       double tol, int early, double *d_w, unsigned char *d_flag, int *d_status)
{
    const size_t lds = (size_t)2 * rows_max * WAVE * sizeof(double);
    if (lds > 160 * 1024) { g_err = "too many facet rows per LP for LDS"; return GCSADMM_ERR_UNSUPPORTED; }
    hipError_t e;
    CK(hipFuncSetAttribute((const void *)ball_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (count > 0)
        hipLaunchKernelGGL(ball_kernel<N>, dim3((unsigned)((count + WAVE - 1) / WAVE)), dim3(WAVE), lds, 0, sc.S, (int)count, d_pa, d_pb,
                           d_x0, rows_max, tol, early, d_w, d_flag, d_status);
    CK(hipGetLastError());
    return GCSADMM_OK;
}
template <int N>
static int launch_bounds(const Scene &sc, const double *d_centers, double *d_lo, double *d_hi, int *d_status)
{
    const int rows_max = sc.maxm + 2 * N;
    const size_t lds = (size_t)2 * rows_max * WAVE * sizeof(double);
    if (lds > 160 * 1024) { g_err = "too many facet rows per LP for LDS"; return GCSADMM_ERR_UNSUPPORTED; }
    hipError_t e;
    CK(hipFuncSetAttribute((const void *)bounds_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long count = (long)sc.S.P * 2 * N;
    if (count > 0)
        hipLaunchKernelGGL(bounds_kernel<N>, dim3((unsigned)((count + WAVE - 1) / WAVE)), dim3(WAVE), lds, 0, sc.S, d_centers, rows_max, d_lo, d_hi, d_status);
    CK(hipGetLastError());
    return GCSADMM_OK;
}

#define DISPATCH_N(n, CALL)                                                                          \
    switch (n) {                                                                                       \
    case 1: { constexpr int NN = 1; rc = CALL; } break;                                                \
    case 2: { constexpr int NN = 2; rc = CALL; } break;                                                \
    case 3: { constexpr int NN = 3; rc = CALL; } break;                                                \
    case 4: { constexpr int NN = 4; rc = CALL; } break;                                                \
    case 5: { constexpr int NN = 5; rc = CALL; } break;                                                \
    case 6: { constexpr int NN = 6; rc = CALL; } break;                                                \
    case 7: { constexpr int NN = 7; rc = CALL; } break;                                                \
    default: { constexpr int NN = 8; rc = CALL; } break;                                               \
    }

} // namespace gcsadmm_lp

using namespace gcsadmm_lp;

extern "C" {

const char *gcsadmm_polytope_last_error(void) { return g_err.c_str(); }

int gcsadmm_polytope_centers(int n, int num_polytopes, const int *poly_ptr, const double *poly_A, const double *poly_b,
                             int device, double *centers, double *radii, int *status)
{
    RestoreDevice restore_device_;
    if (!centers) { g_err = "null output"; return GCSADMM_ERR_BAD_ARG; }
    Scene sc;
    int rc = upload_scene(sc, n, num_polytopes, poly_ptr, poly_A, poly_b, device);
    if (rc != GCSADMM_OK) return rc;
    const int P = num_polytopes;
    DevBuf w, st;
    hipError_t e;
    CK(w.alloc(sizeof(double) * (size_t)P * (n + 1))); CK(st.alloc(sizeof(int) * (size_t)P));
    DISPATCH_N(n, (launch_ball<NN>(sc, P, nullptr, nullptr, nullptr, sc.maxm + 1, 0.0, 0, w.as<double>(), nullptr, st.as<int>())));
    if (rc != GCSADMM_OK) return rc;
    std::vector<double> hw((size_t)P * (n + 1));
    std::vector<int> hs((size_t)P);
    CK(hipMemcpy(hw.data(), w.p, sizeof(double) * hw.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), st.p, sizeof(int) * hs.size(), hipMemcpyDeviceToHost));
    for (int p = 0; p < P; ++p) {
        for (int k = 0; k < n; ++k) centers[(size_t)p * n + k] = hw[(size_t)p * (n + 1) + k];
        if (radii) radii[p] = hw[(size_t)p * (n + 1) + n];
        if (status) status[p] = hs[p];
    }
    return GCSADMM_OK;
}

int gcsadmm_polytope_bounds(int n, int num_polytopes, const int *poly_ptr, const double *poly_A, const double *poly_b,
                            const double *centers, int device, double *lo, double *hi, int *status)
{
    RestoreDevice restore_device_;
    if (!centers || !lo || !hi) { g_err = "null centres or output"; return GCSADMM_ERR_BAD_ARG; }
    Scene sc;
    int rc = upload_scene(sc, n, num_polytopes, poly_ptr, poly_A, poly_b, device);
    if (rc != GCSADMM_OK) return rc;
    const size_t P = (size_t)num_polytopes;
    DevBuf dc, dlo, dhi, st;
    hipError_t e;
    CK(dc.alloc(sizeof(double) * P * n)); CK(dlo.alloc(sizeof(double) * P * n)); CK(dhi.alloc(sizeof(double) * P * n));
    CK(st.alloc(sizeof(int) * P * 2 * n));
    CK(hipMemcpy(dc.p, centers, sizeof(double) * P * n, hipMemcpyHostToDevice));
    DISPATCH_N(n, (launch_bounds<NN>(sc, dc.as<double>(), dlo.as<double>(), dhi.as<double>(), st.as<int>())));
    if (rc != GCSADMM_OK) return rc;
    CK(hipMemcpy(lo, dlo.p, sizeof(double) * P * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hi, dhi.p, sizeof(double) * P * n, hipMemcpyDeviceToHost));
    if (status) CK(hipMemcpy(status, st.p, sizeof(int) * P * 2 * n, hipMemcpyDeviceToHost));
    return GCSADMM_OK;
}

int gcsadmm_polytope_overlaps(int n, int num_polytopes, const int *poly_ptr, const double *poly_A, const double *poly_b,
                              const double *centers, long num_pairs, const int *pair_a, const int *pair_b, double tol,
                              int device, unsigned char *overlap, int *status)
{
    RestoreDevice restore_device_;
    if (num_pairs < 0 || (num_pairs > 0 && (!pair_a || !pair_b || !overlap))) { g_err = "null pair list or output"; return GCSADMM_ERR_BAD_ARG; }
    for (long t = 0; t < num_pairs; ++t)
        if (pair_a[t] < 0 || pair_a[t] >= num_polytopes || pair_b[t] < 0 || pair_b[t] >= num_polytopes) {
            g_err = "pair index out of range"; return GCSADMM_ERR_BAD_ARG;
        }
    Scene sc;
    int rc = upload_scene(sc, n, num_polytopes, poly_ptr, poly_A, poly_b, device);
    if (rc != GCSADMM_OK) return rc;
    const size_t P = (size_t)num_polytopes, T = (size_t)num_pairs;
    DevBuf da, db, dc, df, st;
    hipError_t e;
    CK(da.alloc(sizeof(int) * T)); CK(db.alloc(sizeof(int) * T)); CK(df.alloc(T)); CK(st.alloc(sizeof(int) * T));
    CK(hipMemcpy(da.p, pair_a, sizeof(int) * T, hipMemcpyHostToDevice));
    CK(hipMemcpy(db.p, pair_b, sizeof(int) * T, hipMemcpyHostToDevice));
    if (centers) {
        CK(dc.alloc(sizeof(double) * P * n));
        CK(hipMemcpy(dc.p, centers, sizeof(double) * P * n, hipMemcpyHostToDevice));
    }
    DISPATCH_N(n, (launch_ball<NN>(sc, num_pairs, da.as<int>(), db.as<int>(), centers ? dc.as<double>() : nullptr, 2 * sc.maxm + 1, tol, 1,
                                   nullptr, df.as<unsigned char>(), st.as<int>())));
    if (rc != GCSADMM_OK) return rc;
    CK(hipMemcpy(overlap, df.p, T, hipMemcpyDeviceToHost));
    if (status) CK(hipMemcpy(status, st.p, sizeof(int) * T, hipMemcpyDeviceToHost));
    return GCSADMM_OK;
}

} // extern "C"
