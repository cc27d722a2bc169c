// Counting build of the workgroup vertex program (gcs_admm_amd/csrc/vertex_wg.h): every `double` of the program becomes a
// scalar type that counts its arithmetic, so the number of f64 operations of a vertex step is MEASURED on the algorithm's
// own source instead of modelled (SURVEY.md section 8d asks for an instrumented count behind roofline.achieved_fp).
// MEASUREMENT TOOLING ONLY (bench.py's roofline_fp): host code, never shipped, never timed.
// Conventions: add / sub / mul = 1 flop, fma = 2; divisions and square roots are counted apart (on the device they are
// hardware estimates refined by Newton steps: gcs_math.h); comparisons, min/max, selects, loads and stores count nothing.
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <cstring>
#include <vector>

static long long g_flops = 0, g_div = 0, g_sqrt = 0;

struct counted {
    double v;
    counted() = default;
    constexpr counted(double x) : v(x) {}
    constexpr counted(float x) : v(x) {}
    constexpr counted(int x) : v(x) {}
    explicit operator float() const { return (float)v; }
    explicit operator int() const { return (int)v; }
    explicit operator bool() const { return v != 0.0; }
    counted &operator+=(counted o) { v += o.v; ++g_flops; return *this; }
    counted &operator-=(counted o) { v -= o.v; ++g_flops; return *this; }
    counted &operator*=(counted o) { v *= o.v; ++g_flops; return *this; }
};
static inline counted operator+(counted a, counted b) { ++g_flops; return counted(a.v + b.v); }
static inline counted operator-(counted a, counted b) { ++g_flops; return counted(a.v - b.v); }
static inline counted operator*(counted a, counted b) { ++g_flops; return counted(a.v * b.v); }
static inline counted operator/(counted a, counted b) { ++g_div; return counted(a.v / b.v); }
static inline counted operator-(counted a) { return counted(-a.v); }
static inline bool operator<(counted a, counted b) { return a.v < b.v; }
static inline bool operator>(counted a, counted b) { return a.v > b.v; }
static inline bool operator<=(counted a, counted b) { return a.v <= b.v; }
static inline bool operator>=(counted a, counted b) { return a.v >= b.v; }
static inline bool operator==(counted a, counted b) { return a.v == b.v; }
static inline bool operator!=(counted a, counted b) { return a.v != b.v; }
static inline counted fma(counted a, counted b, counted c) { g_flops += 2; return counted(::fma(a.v, b.v, c.v)); }
static inline counted sqrt(counted a) { ++g_sqrt; return counted(::sqrt(a.v)); }
static inline counted fabs(counted a) { return counted(::fabs(a.v)); }
static inline counted fmin(counted a, counted b) { return counted(::fmin(a.v, b.v)); }
static inline counted fmax(counted a, counted b) { return counted(::fmax(a.v, b.v)); }
// mixed forms the program uses (literal on one side)
#define MIXED(op) \
    static inline counted operator op(counted a, double b) { return a op counted(b); } \
    static inline counted operator op(double a, counted b) { return counted(a) op b; } \
    static inline counted operator op(counted a, int b) { return a op counted((double)b); } \
    static inline counted operator op(int a, counted b) { return counted((double)a) op b; }
MIXED(+) MIXED(-) MIXED(*) MIXED(/)
#undef MIXED
#define MIXEDC(op) \
    static inline bool operator op(counted a, double b) { return a.v op b; } \
    static inline bool operator op(double a, counted b) { return a op b.v; }
MIXEDC(<) MIXEDC(>) MIXEDC(<=) MIXEDC(>=) MIXEDC(==) MIXEDC(!=)
#undef MIXEDC
static inline counted fmin(counted a, double b) { return counted(::fmin(a.v, b)); }
static inline counted fmin(double a, counted b) { return counted(::fmin(a, b.v)); }
static inline counted fmax(counted a, double b) { return counted(::fmax(a.v, b)); }
static inline counted fmax(double a, counted b) { return counted(::fmax(a, b.v)); }

static_assert(sizeof(counted) == sizeof(double), "counted must alias double arrays");

#define double counted
#include "vertex_wg.h"
#undef double

static bool g_box = false;      // count the BOX instantiation (what the library runs when every vertex is a canonical box and n > 2)

template <int N>
static void run_all(const gcs_wg::WgArgs<counted> &a, counted rho, counted mu_scale, int lds, long long *iters)
{
    std::vector<counted> smem(lds);
    for (int w = 0; w < a.n_vtx; ++w) {
        std::fill(smem.begin(), smem.end(), counted(0.0 / 0.0));
        int st = -9, it = 0;
        if (g_box && N > 2) gcs_wg::wg_solve_vertex<N, counted, (N > 2)>(a, a.vtx[w], rho, mu_scale, smem.data(), st, it);
        else gcs_wg::wg_solve_vertex<N, counted, false>(a, a.vtx[w], rho, mu_scale, smem.data(), st, it);
        *iters += it;
    }
}

// counts[0..3] = flops, divisions, square roots, Newton iterations summed over the generic vertices; counts[4] = generic vertices
extern "C" int wg_count_vertex_step(int n, int V, int E, int NI, const int *inc_ptr, const int *inc_edge, const int *inc_out,
                                    const int *poly_ptr, const double *poly_A, const double *poly_b, const double *center,
                                    int src, int dst, const double *zedge, const double *mu, double rho, double mu_scale,
                                    double eps_edge, double ipm_tol, int ipm_max_iter, long long *counts)
{
    if (n != 2 && n != 3 && n != 6) return 1;
    std::vector<int> deg_in(V, 0), vtx;
    int lds = 0;
    // the same rule as gcsadmm_create (canonical_box.h): BOX instantiation when every counted vertex is a canonical box, n > 2
    g_box = n > 2;
    for (int v = 0; v < V && g_box; ++v) {
        const int m = poly_ptr[v + 1] - poly_ptr[v];
        if (m != 2 * n) { g_box = false; break; }
        for (int j = 0; j < m && g_box; ++j)
            for (int k = 0; k < n; ++k)
                if (poly_A[((size_t)poly_ptr[v] + j) * n + k] != ((j % n) == k ? (j < n ? 1.0 : -1.0) : 0.0)) { g_box = false; break; }
    }
    for (int v = 0; v < V; ++v) {
        for (int k = inc_ptr[v]; k < inc_ptr[v + 1]; ++k) deg_in[v] += !inc_out[k];
        const int d = inc_ptr[v + 1] - inc_ptr[v];
        if (!(v == src || v == dst || deg_in[v] == 0 || d - deg_in[v] == 0)) {
            vtx.push_back(v);
            lds = std::max(lds, gcs_wg::wg_lds_doubles_n(n, d + 1, poly_ptr[v + 1] - poly_ptr[v], g_box));
        }
    }
    std::vector<counted> bc(poly_ptr[V]), A((size_t)poly_ptr[V] * n), cen((size_t)V * n), ze((size_t)(2 * n + 1) * E), m_((size_t)(2 * n + 1) * NI);
    for (int v = 0; v < V; ++v)
        for (int j = poly_ptr[v]; j < poly_ptr[v + 1]; ++j) {
            double s = poly_b[j];
            for (int k = 0; k < n; ++k) s -= poly_A[(size_t)j * n + k] * center[(size_t)v * n + k];
            bc[j] = counted(s);
        }
    for (size_t i = 0; i < A.size(); ++i) A[i] = counted(poly_A[i]);
    for (size_t i = 0; i < cen.size(); ++i) cen[i] = counted(center[i]);
    for (size_t i = 0; i < ze.size(); ++i) ze[i] = counted(zedge[i]);
    for (size_t i = 0; i < m_.size(); ++i) m_[i] = counted(mu[i]);
    std::vector<counted> copy((size_t)(2 * n + 1) * NI), xv((size_t)V * 2 * n), zv((size_t)V * 2 * n), yv(V);
    int counters[2] = {0, 0};
    gcs_wg::WgArgs<counted> a;
    a.n_vtx = (int)vtx.size(); a.vtx = vtx.data();
    a.inc_ptr = inc_ptr; a.deg_in = deg_in.data(); a.inc_edge = inc_edge; a.poly_ptr = poly_ptr;
    a.poly_A = A.data(); a.poly_bc = bc.data(); a.center = cen.data(); a.E = E; a.NI = NI;
    a.zedge = ze.data(); a.mu = m_.data(); a.copy = copy.data(); a.xv = xv.data(); a.zv = zv.data(); a.yv = yv.data();
    a.counters = counters; a.eps_edge = counted(eps_edge); a.ipm_tol = counted(ipm_tol); a.ipm_max_iter = ipm_max_iter;
    g_flops = g_div = g_sqrt = 0;
    long long iters = 0;
    if (n == 2) run_all<2>(a, counted(rho), counted(mu_scale), lds, &iters);
    else if (n == 3) run_all<3>(a, counted(rho), counted(mu_scale), lds, &iters);
    else run_all<6>(a, counted(rho), counted(mu_scale), lds, &iters);
    counts[0] = g_flops; counts[1] = g_div; counts[2] = g_sqrt; counts[3] = iters; counts[4] = (long long)vtx.size();
    return 0;
}
