// vertex_wg.hip -- gfx950 kernel of the workgroup-cooperative vertex program (vertex_wg.h): one 256-thread workgroup
// per generic vertex, any degree and facet count that fits the CU's 160 KB of LDS.  Trailing workgroups of the launch take the
// closed-form vertices (special_vertex.h).  This object instantiates the program for n = 2, 3, 6 and dispatches; n = 1, 4, 5 are in
// vertex_wg_dims.hip (same templates, vertex_wg_kernel.h).
// Replaces admm_solver_v3.py:469-540 (one MOSEK solve per vertex through SolveInParallel) for the vertices routed here
// by gcsadmm_create: small graphs, n != 2, degree > 63.
// The object is built twice (gcs_admm_amd/build.py): with 256 threads per workgroup, and with 512 for launches of at most one workgroup
// per CU (-DGCS_WG_THREADS=512 -Dgcs_wg=gcs_wg_t512 -D'GCS_WG_SYM(name)=name##_t512': own namespace, own entry points; gcsadmm.hip
// chooses at create).
#include "vertex_wg_kernel.h"

using namespace gcsadmm_k;

#ifndef GCS_WG_SYM
#define GCS_WG_SYM(name) name
#endif

// n = 1, 4, 5, 7, 8 (vertex_wg_dims.hip, the object of the same thread count)
hipError_t GCS_WG_SYM(gcsadmm_wg_set_lds_dims)(int n, int dtype, int lds_bytes);
void GCS_WG_SYM(gcsadmm_wg_launch_dims)(const WgLaunchDesc &d, hipStream_t s);
void GCS_WG_SYM(gcsadmm_wg_launch_prox_dims)(const WgLaunchDesc &d, const double *q, const double *c, int src, int dst, hipStream_t s);


int GCS_WG_SYM(gcsadmm_wg_lds_bytes)(int n, int units, int facets, bool box) { return 8 * gcs_wg::wg_lds_doubles_n(n, units, facets, box); }
bool GCS_WG_SYM(gcsadmm_wg_has_box)(int n) { return gcs_wg::wg_has_box(n); }

hipError_t GCS_WG_SYM(gcsadmm_wg_set_lds)(int n, int dtype, int lds_bytes)
{
    const bool f64 = dtype == GCSADMM_F64;
    if (n == 2) return f64 ? set_lds<2, double>(lds_bytes) : set_lds<2, float>(lds_bytes);
    if (n == 3) return f64 ? set_lds<3, double>(lds_bytes) : set_lds<3, float>(lds_bytes);
    if (n == 6) return f64 ? set_lds<6, double>(lds_bytes) : set_lds<6, float>(lds_bytes);
    return GCS_WG_SYM(gcsadmm_wg_set_lds_dims)(n, dtype, lds_bytes);
}

void GCS_WG_SYM(gcsadmm_wg_launch)(const WgLaunchDesc &d, hipStream_t s)
{
    const bool f64 = d.dtype == GCSADMM_F64;
    if (d.n == 2) { if (f64) launch<2, double>(d, s); else launch<2, float>(d, s); }
    else if (d.n == 3) { if (f64) launch<3, double>(d, s); else launch<3, float>(d, s); }
    else if (d.n == 6) { if (f64) launch<6, double>(d, s); else launch<6, float>(d, s); }
    else GCS_WG_SYM(gcsadmm_wg_launch_dims)(d, s);
}

void GCS_WG_SYM(gcsadmm_wg_launch_prox)(const WgLaunchDesc &d, const double *q, const double *c, int src, int dst, hipStream_t s)
{
    if (d.n == 2) launch_prox<2>(d, q, c, src, dst, s);
    else if (d.n == 3) launch_prox<3>(d, q, c, src, dst, s);
    else if (d.n == 6) launch_prox<6>(d, q, c, src, dst, s);
    else GCS_WG_SYM(gcsadmm_wg_launch_prox_dims)(d, q, c, src, dst, s);
}

#ifdef GCS_WG_TIMING
extern "C" int gcsadmm_debug_wg_cycles(unsigned long long *cycles64, unsigned long long *counts64)
{
    int e = (int)hipMemcpyFromSymbol(cycles64, HIP_SYMBOL(gcs_wg::g_wg_cycles), 64 * sizeof(unsigned long long));
    if (e == 0) e = (int)hipMemcpyFromSymbol(counts64, HIP_SYMBOL(gcs_wg::g_wg_counts), 64 * sizeof(unsigned long long));
    return e;
}
extern "C" int gcsadmm_debug_wg_wave_cycles(unsigned long long *cycles512)
{
    return (int)hipMemcpyFromSymbol(cycles512, HIP_SYMBOL(gcs_wg::g_wg_wave_cycles), 512 * sizeof(unsigned long long));
}
#endif
#ifdef GCS_WG_BLOCKTIME
extern "C" int gcsadmm_debug_wg_blocks(unsigned long long *ticks64, unsigned long long *iters64)
{
    int e = (int)hipMemcpyFromSymbol(ticks64, HIP_SYMBOL(g_wg_block_ticks), 64 * sizeof(unsigned long long));
    if (e == 0) e = (int)hipMemcpyFromSymbol(iters64, HIP_SYMBOL(g_wg_block_iters), 64 * sizeof(unsigned long long));
    return e;
}
#endif
