// HOST build of the workgroup-cooperative vertex program (gcs_admm_amd/csrc/vertex_wg.h).
// DEBUG / TEST HARNESS ONLY: with WG_FOR a plain loop and WG_SYNC a no-op each parallel region runs its tasks
// serially, which equals the GPU execution as long as the tasks of a region are independent.  Built twice
// (tasks in ascending order / -DGCS_WG_REVERSE descending order): equal results are evidence of that independence.
// Unwritten LDS is poisoned with NaN.  Never shipped, never timed; the product has no path into it.
#include <algorithm>
#include <cstring>
#include <vector>

#include "vertex_wg.h"

#ifdef GCS_WG_REVERSE
#define EMU_FN wg_emu_vertex_step_rev
#define EMU_BOX wg_emu_set_box_rev
#define EMU_LDS wg_emu_lds_doubles_rev
#define EMU_WARM wg_emu_set_warm_rev
#else
#define EMU_WARM wg_emu_set_warm
#define EMU_FN wg_emu_vertex_step
#define EMU_BOX wg_emu_set_box
#define EMU_LDS wg_emu_lds_doubles
#endif

static bool g_emu_box = false;      // set per call: run the BOX instantiation (the caller vouches for canonical boxes)
static double *g_emu_warm = nullptr;                // warm-start records of the following steps (warm_start.h); null = cold solves
static const long long *g_emu_warm_ptr = nullptr;

template <int N>
static void run_all(const gcs_wg::WgArgs<double> &a, double rho, double mu_scale, int lds, int *status, int *iters)
{
    std::vector<double> smem(lds);
    for (int w = 0; w < a.n_vtx; ++w) {
        std::fill(smem.begin(), smem.end(), 0.0 / 0.0);
        int st = -9, it = 0;
        if (g_emu_box && (N == 3 || N == 6)) gcs_wg::wg_solve_vertex<N, double, (N == 3 || N == 6)>(a, a.vtx[w], rho, mu_scale, smem.data(), st, it);
        else gcs_wg::wg_solve_vertex<N, double, false>(a, a.vtx[w], rho, mu_scale, smem.data(), st, it);
        status[a.vtx[w]] = st; iters[a.vtx[w]] = it;
        a.counters[0] += st != 0; a.counters[1] += it;
    }
}

extern "C" int EMU_LDS(int n, int U, int m) { return gcs_wg::wg_lds_doubles_n(n, U, m, g_emu_box); }      // (layout of the mode set by EMU_BOX)
// warm-start records for the following vertex steps: warm + warm_ptr[v], wg_emu_warm_doubles(n, m, d) doubles each, zeroed by the caller
extern "C" void EMU_WARM(double *warm, const long long *warm_ptr) { g_emu_warm = warm; g_emu_warm_ptr = warm_ptr; }
#ifndef GCS_WG_REVERSE
extern "C" long long wg_emu_warm_doubles(int n, int m, int d) { return gcs_ws::warm_record_doubles(n, m, d); }
#endif
// 1: the following steps run the BOX instantiation of the program (every polytope must be a canonical box: canonical_box.h)
extern "C" void EMU_BOX(int on) { g_emu_box = on != 0; }

// one vertex step over the generic vertices of a graph; special vertices (s, t, no-flow) are left untouched
extern "C" int EMU_FN(int n, int V, int E, int NI, const int *inc_ptr, const int *inc_edge, const int *inc_out,
                      const int *poly_ptr, const double *poly_A, const double *poly_b, const double *center,
                      int src, int dst, const double *zedge, const double *mu, double rho, double mu_scale,
                      double eps_edge, double ipm_tol, int ipm_max_iter, double *copy, double *xv, double *zv,
                      double *yv, int *counters, int *is_generic, int *status, int *iters)
{
    if (n < 1 || n > gcs_wg::WG_MAX_N) return 1;
    std::vector<int> deg_in(V, 0), vtx;
    int lds = 0;
    for (int v = 0; v < V; ++v) {
        for (int k = inc_ptr[v]; k < inc_ptr[v + 1]; ++k) deg_in[v] += !inc_out[k];
        const int d = inc_ptr[v + 1] - inc_ptr[v];
        is_generic[v] = !(v == src || v == dst || deg_in[v] == 0 || d - deg_in[v] == 0);
        status[v] = 0; iters[v] = 0;
        if (is_generic[v]) {
            vtx.push_back(v);
            lds = std::max(lds, gcs_wg::wg_lds_doubles_n(n, d + 1, poly_ptr[v + 1] - poly_ptr[v], g_emu_box));
        }
    }
    std::vector<double> bc(poly_ptr[V]);
    for (int v = 0; v < V; ++v)
        for (int j = poly_ptr[v]; j < poly_ptr[v + 1]; ++j) {
            double s = poly_b[j];
            for (int k = 0; k < n; ++k) s -= poly_A[(size_t)j * n + k] * center[(size_t)v * n + k];
            bc[j] = s;
        }
    gcs_wg::WgArgs<double> a;
    a.n_vtx = (int)vtx.size(); a.vtx = vtx.data();
    a.inc_ptr = inc_ptr; a.deg_in = deg_in.data(); a.inc_edge = inc_edge; a.poly_ptr = poly_ptr;
    a.poly_A = poly_A; a.poly_bc = bc.data(); a.center = center; a.E = E; a.NI = NI;
    a.zedge = zedge; a.mu = mu; a.copy = copy; a.xv = xv; a.zv = zv; a.yv = yv; a.counters = counters;
    a.eps_edge = eps_edge; a.ipm_tol = ipm_tol; a.ipm_max_iter = ipm_max_iter;
    a.warm = g_emu_warm; a.warm_ptr = g_emu_warm_ptr;
    switch (n) {
    case 1: run_all<1>(a, rho, mu_scale, lds, status, iters); break;
    case 2: run_all<2>(a, rho, mu_scale, lds, status, iters); break;
    case 3: run_all<3>(a, rho, mu_scale, lds, status, iters); break;
    case 4: run_all<4>(a, rho, mu_scale, lds, status, iters); break;
    case 5: run_all<5>(a, rho, mu_scale, lds, status, iters); break;
    case 6: run_all<6>(a, rho, mu_scale, lds, status, iters); break;
    case 7: run_all<7>(a, rho, mu_scale, lds, status, iters); break;
    default: run_all<8>(a, rho, mu_scale, lds, status, iters);
    }
    return 0;
}

#ifdef GCS_WG_REVERSE
#define EMU_PROX wg_emu_vertex_prox_rev
#else
#define EMU_PROX wg_emu_vertex_prox
#endif
// the PROX configuration of the same program (x-update of the vertex-edge splits, admm_solver_v1.py:334-383):
// every vertex except the two terminals; q, c are [V][4n+1] (order x, z, y)
extern "C" int EMU_PROX(int n, int V, const int *poly_ptr, const double *poly_A, const double *poly_b, const double *center,
                        int src, int dst, const double *q, const double *c, double ipm_tol, int ipm_max_iter,
                        double *xv, double *zv, double *yv, int *counters, int *status, int *iters)
{
    if (n != 2 && n != 3 && n != 6) return 1;
    std::vector<int> vtx, zero(V + 1, 0);
    int lds = 0;
    for (int v = 0; v < V; ++v) {
        status[v] = 0; iters[v] = 0;
        if (v == src || v == dst) continue;
        vtx.push_back(v);
        lds = std::max(lds, gcs_wg::wg_lds_doubles_n(n, 1, poly_ptr[v + 1] - poly_ptr[v], g_emu_box));
    }
    std::vector<double> bc(poly_ptr[V]);
    for (int v = 0; v < V; ++v)
        for (int j = poly_ptr[v]; j < poly_ptr[v + 1]; ++j) {
            double s = poly_b[j];
            for (int k = 0; k < n; ++k) s -= poly_A[(size_t)j * n + k] * center[(size_t)v * n + k];
            bc[j] = s;
        }
    gcs_wg::WgArgs<double> a{};
    a.n_vtx = (int)vtx.size(); a.vtx = vtx.data();
    a.inc_ptr = zero.data(); a.deg_in = zero.data(); a.inc_edge = zero.data(); a.poly_ptr = poly_ptr;
    a.poly_A = poly_A; a.poly_bc = bc.data(); a.center = center; a.E = 0; a.NI = 0;
    a.zedge = nullptr; a.mu = nullptr; a.copy = nullptr; a.xv = xv; a.zv = zv; a.yv = yv; a.counters = counters;
    a.eps_edge = 0.0; a.ipm_tol = ipm_tol; a.ipm_max_iter = ipm_max_iter; a.prox_q = q; a.prox_c = c;
    if (n == 2) run_all<2>(a, 1.0, 1.0, lds, status, iters);
    else if (n == 3) run_all<3>(a, 1.0, 1.0, lds, status, iters);
    else run_all<6>(a, 1.0, 1.0, lds, status, iters);
    return 0;
}
