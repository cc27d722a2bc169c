// vertex_wg_dims.hip -- the workgroup-cooperative vertex program (vertex_wg.h, vertex_wg_kernel.h) instantiated for the space
// dimensions BASELINE.json does not name: n = 1, 4, 5, 7, 8 (7: the configuration space of a seven-joint arm).  The reference's sub-problem takes
// any n (admm_solver_v3.py:363-377); the program is the same template.  Generic instantiation only (the BOX one exists for the tuned dimensions 3 and 6).
// Built twice like vertex_wg.hip (256 / 512 threads per workgroup: gcs_admm_amd/build.py).
#include "vertex_wg_kernel.h"

using namespace gcsadmm_k;

#ifndef GCS_WG_SYM
#define GCS_WG_SYM(name) name
#endif

hipError_t GCS_WG_SYM(gcsadmm_wg_set_lds_dims)(int n, int dtype, int lds_bytes)
{
    const bool f64 = dtype == GCSADMM_F64;
    if (n == 1) return f64 ? set_lds<1, double>(lds_bytes) : set_lds<1, float>(lds_bytes);
    if (n == 4) return f64 ? set_lds<4, double>(lds_bytes) : set_lds<4, float>(lds_bytes);
    if (n == 5) return f64 ? set_lds<5, double>(lds_bytes) : set_lds<5, float>(lds_bytes);
    if (n == 7) return f64 ? set_lds<7, double>(lds_bytes) : set_lds<7, float>(lds_bytes);
    if (n == 8) return f64 ? set_lds<8, double>(lds_bytes) : set_lds<8, float>(lds_bytes);
    return hipErrorInvalidValue;
}

void GCS_WG_SYM(gcsadmm_wg_launch_dims)(const WgLaunchDesc &d, hipStream_t s)
{
    const bool f64 = d.dtype == GCSADMM_F64;
    if (d.n == 1) { if (f64) launch<1, double>(d, s); else launch<1, float>(d, s); }
    else if (d.n == 4) { if (f64) launch<4, double>(d, s); else launch<4, float>(d, s); }
    else if (d.n == 5) { if (f64) launch<5, double>(d, s); else launch<5, float>(d, s); }
    else if (d.n == 7) { if (f64) launch<7, double>(d, s); else launch<7, float>(d, s); }
    else if (d.n == 8) { if (f64) launch<8, double>(d, s); else launch<8, float>(d, s); }
}

void GCS_WG_SYM(gcsadmm_wg_launch_prox_dims)(const WgLaunchDesc &d, const double *q, const double *c, int src, int dst, hipStream_t s)
{
    if (d.n == 1) launch_prox<1>(d, q, c, src, dst, s);
    else if (d.n == 4) launch_prox<4>(d, q, c, src, dst, s);
    else if (d.n == 5) launch_prox<5>(d, q, c, src, dst, s);
    else if (d.n == 7) launch_prox<7>(d, q, c, src, dst, s);
    else if (d.n == 8) launch_prox<8>(d, q, c, src, dst, s);
}
