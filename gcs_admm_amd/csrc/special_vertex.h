// special_vertex.h -- closed-form x-update of the vertices that need no interior-point solve (shared by the wavefront
// kernel, vertex_kernel.h, and the workgroup kernel, vertex_wg.hip; they ride in the trailing workgroups of a launch).
#pragma once
#include <hip/hip_runtime.h>

namespace gcsadmm_k {

constexpr int MAX_SPECIAL_DEG = 256;

// -------------------------------------------------------------------------------------------------
// special vertices: s / t are points (utils.py:12-28, boxes of half-width 1e-6) -> the sub-problem
// collapses to a separable quadratic over the simplex of the live side; a vertex with no incoming or
// no outgoing edge carries no flow.  One thread per vertex.
// -------------------------------------------------------------------------------------------------
template <class T> struct SpecialArgs {
    int count;
    const int *vtx;     // vertex ids
    const int *kind;    // 1 = source, 2 = target, 0 = no-flow
    const int *inc_ptr, *deg_in, *inc_edge;
    const double *center;
    int E, NI;
    const T *zedge, *mu;
    T *copy;
    double *xv, *zv, *yv;
    double eps_edge;
    int edge_major = 0;     // 1: state columns numbered by edge (tail side e, head side E + e) instead of by incidence
};

// vals / u: work arrays of MAX_SPECIAL_DEG doubles each, used by the source and the target only
template <int N, class T>
__device__ void special_body(const SpecialArgs<T> &a, int i, double rho, double mu_scale, double *vals, double *u)
{
    const int v = a.vtx[i], kind = a.kind[i];
    const int lo = a.inc_ptr[v], d = a.inc_ptr[v + 1] - lo, d_in = a.deg_in[v];
    double cen[N];
#pragma unroll
    for (int k = 0; k < N; ++k) cen[k] = a.center[(size_t)v * N + k];
    auto target = [&](int w, int k) -> double {
        const int e = a.inc_edge[lo + k], inc = a.edge_major ? e + (k >= d_in ? 0 : a.E) : lo + k;
        return (double)a.zedge[(size_t)w * a.E + e] - mu_scale * (double)a.mu[(size_t)w * a.NI + inc];
    };
    const bool is_src = kind == 1, is_dst = kind == 2;
    const int live_lo = is_src ? d_in : 0, live_hi = is_src ? d : (is_dst ? d_in : 0);
    const int na = live_hi - live_lo;
    double tau = 0.0;
    if (na > 0) {
        double pp = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) pp += cen[k] * cen[k];
        const double aq = is_src ? 2 * pp + 1 : pp + 1;
        for (int e = live_lo; e < live_hi; ++e) {
            double cc = target(2 * N, e);
#pragma unroll
            for (int k = 0; k < N; ++k) cc += cen[k] * (is_src ? target(k, e) + target(N + k, e) : target(N + k, e));
            vals[e - live_lo] = (cc - a.eps_edge / rho) / aq;
        }
        // threshold of the Euclidean projection onto the simplex: sort descending (insertion), scan
        for (int q = 0; q < na; ++q) u[q] = vals[q];
        for (int q = 1; q < na; ++q) {
            const double x = u[q];
            int j = q - 1;
            while (j >= 0 && u[j] < x) { u[j + 1] = u[j]; --j; }
            u[j + 1] = x;
        }
        double css = 0;
        for (int k = 0; k < na; ++k) {
            css += u[k];
            if (u[k] * (k + 1) > css - 1.0) tau = (css - 1.0) / (k + 1);
        }
    }
    for (int e = 0; e < d; ++e) {
        const bool live = e >= live_lo && e < live_hi;
        double ye = 0.0;
        if (live) { ye = vals[e - live_lo] - tau; ye = ye > 0 ? ye : 0.0; }
        const bool outgoing = e >= d_in;
        const int inc = a.edge_major ? a.inc_edge[lo + e] + (outgoing ? 0 : a.E) : lo + e;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double yc = (kind != 0) ? ye * cen[k] : 0.0;
            a.copy[(size_t)k * a.NI + inc] = (T)(outgoing ? yc : target(k, e));
            a.copy[(size_t)(N + k) * a.NI + inc] = (T)yc;
        }
        a.copy[(size_t)(2 * N) * a.NI + inc] = (T)ye;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        a.xv[(size_t)v * 2 * N + k] = a.xv[(size_t)v * 2 * N + N + k] = cen[k];
        a.zv[(size_t)v * 2 * N + k] = a.zv[(size_t)v * 2 * N + N + k] = (kind != 0) ? cen[k] : 0.0;
    }
    a.yv[v] = (kind != 0) ? 1.0 : 0.0;
}


} // namespace gcsadmm_k
