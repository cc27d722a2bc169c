// Lock-step HOST emulation of the vertex-step wavefront program (gcs_admm_amd/csrc/vertex_program.h).
// DEBUG / TEST HARNESS ONLY: lets the lane program be exercised (and run under sanitizers) on a
// machine without a GPU.  Each barrier-separated phase is executed for lanes 0..63 in turn.  It is
// never shipped, never timed, and the product has no path into it.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "vertex_program.h"


// the program exists in two instantiations (vertex_program.h); emulate both
#define EMU_NAME emu_vertex_step
namespace emu_generic {
using namespace gcs;


template <int N> struct CpuExec {
    Lane<N> Ls[WAVE];
    template <class F> void each(F &&f) { for (int l = 0; l < WAVE; ++l) f(Ls[l], l); }
    template <class P> bool all(P &&p) { for (int l = 0; l < WAVE; ++l) if (!p(Ls[l])) return false; return true; }
    void count(int *c, int fails, int iters) { c[0] += fails; c[1] += iters; }
    template <class F> int wave_max(F &&f) { int m = 0; for (int l = 0; l < WAVE; ++l) m = std::max(m, f(Ls[l])); return m; }
    // lanes run in increasing order inside a phase: the head lane of a segment initialises the sum, the
    // following lanes of the segment accumulate (the GPU does the same reduction as a shuffle tree)
    template <int CNT> void seg_reduce(Lane<N> &L, double (&v)[CNT], double *sin, double *sout, int special, int op,
                                       bool contributes, int, bool)
    {
        if (!contributes) return;
        double *dst = L.out ? sout : sin;
        for (int k = 0; k < CNT; ++k) {
            if (L.seg_head) dst[k] = v[k];
            else if (k == special && op == 1) dst[k] = fmin(dst[k], v[k]);
            else if (k == special && op == 2) dst[k] = fmax(dst[k], v[k]);
            else dst[k] += v[k];
        }
    }
};

extern "C" int EMU_NAME(int n, int V, int E, int NI, const int *inc_ptr, const int *inc_edge, const int *inc_out,
                               const int *poly_ptr, const double *poly_A, const double *poly_b, const double *center,
                               int src, int dst, const double *zedge, const double *mu, double rho, double mu_scale,
                               double eps_edge, double ipm_tol, int ipm_max_iter, double *copy, double *xv, double *zv,
                               double *yv, int *counters, int *is_generic)
{
    if (n != 2 && n != 3 && n != 6) return 1;
    std::vector<int> deg_in(V, 0);
    int MM = 1;
    for (int v = 0; v < V; ++v) {
        for (int k = inc_ptr[v]; k < inc_ptr[v + 1]; ++k) deg_in[v] += !inc_out[k];
        MM = std::max(MM, poly_ptr[v + 1] - poly_ptr[v]);
    }
    std::vector<double> bc(poly_ptr[V]);
    for (int v = 0; v < V; ++v)
        for (int j = poly_ptr[v]; j < poly_ptr[v + 1]; ++j) {
            double s = poly_b[j];
            for (int k = 0; k < n; ++k) s -= poly_A[(size_t)j * n + k] * center[(size_t)v * n + k];
            bc[j] = s;
        }
    std::vector<int> wave_slot_ptr{0}, wave_vtx;
    const int align_rows = getenv("GCS_EMU_ALIGN") ? atoi(getenv("GCS_EMU_ALIGN")) : 1;   // exercise the aligned placement by default
    const int STORE_DL = getenv("GCS_EMU_STORE_DL") ? atoi(getenv("GCS_EMU_STORE_DL")) : 1;   // and the stored dual directions
    int lanes = 0, slots = 0;
    for (int v = 0; v < V; ++v) {
        const int d = inc_ptr[v + 1] - inc_ptr[v], din = deg_in[v];
        is_generic[v] = !(v == src || v == dst || din == 0 || d - din == 0);
        if (!is_generic[v]) continue;
        if (d + 1 > WAVE) return 2;
        int b = group_base(lanes, d, din, align_rows);
        if (b < 0 || slots + 1 > MAX_SLOTS) { wave_slot_ptr.push_back((int)wave_vtx.size()); slots = 0; b = group_base(0, d, din, align_rows); }
        wave_vtx.push_back(v); lanes = b + d + 1; slots += 1;
    }
    if ((int)wave_vtx.size() > wave_slot_ptr.back()) wave_slot_ptr.push_back((int)wave_vtx.size());
    const int n_waves = (int)wave_slot_ptr.size() - 1;
    VertexArgs<double> a;
    a.n_waves = n_waves; a.wave_slot_ptr = wave_slot_ptr.data(); a.wave_vtx = wave_vtx.data(); a.align_rows = align_rows;
    a.inc_ptr = inc_ptr; a.deg_in = deg_in.data(); a.inc_edge = inc_edge; a.poly_ptr = poly_ptr;
    a.poly_A = poly_A; a.poly_bc = bc.data(); a.center = center; a.E = E; a.NI = NI; a.MM = MM;
    a.zedge = zedge; a.mu = mu; a.copy = copy; a.xv = xv; a.zv = zv; a.yv = yv; a.counters = counters;
    a.eps_edge = eps_edge; a.ipm_tol = ipm_tol; a.ipm_max_iter = ipm_max_iter;
    std::vector<double> smem(lds_doubles(n, MM, MAX_SLOTS, STORE_DL));
    auto run_all = [&](auto *ex, auto ntag) {
        constexpr int NN = decltype(ntag)::value;
        for (int w = 0; w < n_waves; ++w) {
            std::fill(smem.begin(), smem.end(), 0.0 / 0.0);   // poison: reads of unwritten LDS show up as NaN
            WaveShared S;
            wave_shared_init(S, smem.data(), n, MM, STORE_DL);
            if (STORE_DL) run_vertex_program<NN, double, 1>(*ex, w, a, S, rho, mu_scale);
            else run_vertex_program<NN, double, 0>(*ex, w, a, S, rho, mu_scale);
        }
        delete ex;
    };
    if (n == 2) run_all(new CpuExec<2>(), std::integral_constant<int, 2>());
    else if (n == 3) run_all(new CpuExec<3>(), std::integral_constant<int, 3>());
    else run_all(new CpuExec<6>(), std::integral_constant<int, 6>());
    return 0;
}

// LDS slot size in doubles (layout rule checks)
extern "C" int emu_slot_size(int n, int mm)
{
    return n == 2 ? SlotLayout<2>(mm).SIZE : (n == 3 ? SlotLayout<3>(mm).SIZE : SlotLayout<6>(mm).SIZE);
}
extern "C" int emu_group_base(int cur, int d, int d_in, int align) { return group_base(cur, d, d_in, align); }
} // namespace emu_generic
#undef EMU_NAME
#define EMU_NAME emu_vertex_step_m4
namespace emu_m4 {
using namespace gcs_m4;


template <int N> struct CpuExec {
    Lane<N> Ls[WAVE];
    template <class F> void each(F &&f) { for (int l = 0; l < WAVE; ++l) f(Ls[l], l); }
    template <class P> bool all(P &&p) { for (int l = 0; l < WAVE; ++l) if (!p(Ls[l])) return false; return true; }
    void count(int *c, int fails, int iters) { c[0] += fails; c[1] += iters; }
    template <class F> int wave_max(F &&f) { int m = 0; for (int l = 0; l < WAVE; ++l) m = std::max(m, f(Ls[l])); return m; }
    // lanes run in increasing order inside a phase: the head lane of a segment initialises the sum, the
    // following lanes of the segment accumulate (the GPU does the same reduction as a shuffle tree)
    template <int CNT> void seg_reduce(Lane<N> &L, double (&v)[CNT], double *sin, double *sout, int special, int op,
                                       bool contributes, int, bool)
    {
        if (!contributes) return;
        double *dst = L.out ? sout : sin;
        for (int k = 0; k < CNT; ++k) {
            if (L.seg_head) dst[k] = v[k];
            else if (k == special && op == 1) dst[k] = fmin(dst[k], v[k]);
            else if (k == special && op == 2) dst[k] = fmax(dst[k], v[k]);
            else dst[k] += v[k];
        }
    }
};

extern "C" int EMU_NAME(int n, int V, int E, int NI, const int *inc_ptr, const int *inc_edge, const int *inc_out,
                               const int *poly_ptr, const double *poly_A, const double *poly_b, const double *center,
                               int src, int dst, const double *zedge, const double *mu, double rho, double mu_scale,
                               double eps_edge, double ipm_tol, int ipm_max_iter, double *copy, double *xv, double *zv,
                               double *yv, int *counters, int *is_generic)
{
    if (n != 2 && n != 3 && n != 6) return 1;
    std::vector<int> deg_in(V, 0);
    int MM = 1;
    for (int v = 0; v < V; ++v) {
        for (int k = inc_ptr[v]; k < inc_ptr[v + 1]; ++k) deg_in[v] += !inc_out[k];
        MM = std::max(MM, poly_ptr[v + 1] - poly_ptr[v]);
    }
    std::vector<double> bc(poly_ptr[V]);
    for (int v = 0; v < V; ++v)
        for (int j = poly_ptr[v]; j < poly_ptr[v + 1]; ++j) {
            double s = poly_b[j];
            for (int k = 0; k < n; ++k) s -= poly_A[(size_t)j * n + k] * center[(size_t)v * n + k];
            bc[j] = s;
        }
    std::vector<int> wave_slot_ptr{0}, wave_vtx;
    const int align_rows = getenv("GCS_EMU_ALIGN") ? atoi(getenv("GCS_EMU_ALIGN")) : 1;   // exercise the aligned placement by default
    const int STORE_DL = getenv("GCS_EMU_STORE_DL") ? atoi(getenv("GCS_EMU_STORE_DL")) : 1;   // and the stored dual directions
    int lanes = 0, slots = 0;
    for (int v = 0; v < V; ++v) {
        const int d = inc_ptr[v + 1] - inc_ptr[v], din = deg_in[v];
        is_generic[v] = !(v == src || v == dst || din == 0 || d - din == 0);
        if (!is_generic[v]) continue;
        if (d + 1 > WAVE) return 2;
        int b = group_base(lanes, d, din, align_rows);
        if (b < 0 || slots + 1 > MAX_SLOTS) { wave_slot_ptr.push_back((int)wave_vtx.size()); slots = 0; b = group_base(0, d, din, align_rows); }
        wave_vtx.push_back(v); lanes = b + d + 1; slots += 1;
    }
    if ((int)wave_vtx.size() > wave_slot_ptr.back()) wave_slot_ptr.push_back((int)wave_vtx.size());
    const int n_waves = (int)wave_slot_ptr.size() - 1;
    VertexArgs<double> a;
    a.n_waves = n_waves; a.wave_slot_ptr = wave_slot_ptr.data(); a.wave_vtx = wave_vtx.data(); a.align_rows = align_rows;
    a.inc_ptr = inc_ptr; a.deg_in = deg_in.data(); a.inc_edge = inc_edge; a.poly_ptr = poly_ptr;
    a.poly_A = poly_A; a.poly_bc = bc.data(); a.center = center; a.E = E; a.NI = NI; a.MM = MM;
    a.zedge = zedge; a.mu = mu; a.copy = copy; a.xv = xv; a.zv = zv; a.yv = yv; a.counters = counters;
    a.eps_edge = eps_edge; a.ipm_tol = ipm_tol; a.ipm_max_iter = ipm_max_iter;
    std::vector<double> smem(lds_doubles(n, MM, MAX_SLOTS, STORE_DL));
    auto run_all = [&](auto *ex, auto ntag) {
        constexpr int NN = decltype(ntag)::value;
        for (int w = 0; w < n_waves; ++w) {
            std::fill(smem.begin(), smem.end(), 0.0 / 0.0);   // poison: reads of unwritten LDS show up as NaN
            WaveShared S;
            wave_shared_init(S, smem.data(), n, MM, STORE_DL);
            if (STORE_DL) run_vertex_program<NN, double, 1>(*ex, w, a, S, rho, mu_scale);
            else run_vertex_program<NN, double, 0>(*ex, w, a, S, rho, mu_scale);
        }
        delete ex;
    };
    if (n == 2) run_all(new CpuExec<2>(), std::integral_constant<int, 2>());
    else if (n == 3) run_all(new CpuExec<3>(), std::integral_constant<int, 3>());
    else run_all(new CpuExec<6>(), std::integral_constant<int, 6>());
    return 0;
}

} // namespace emu_m4
