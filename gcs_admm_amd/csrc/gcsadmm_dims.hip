// gcsadmm_dims.hip -- the vertex-step kernels for space dimensions 3 and 6 (same wavefront program as n = 2,
// see vertex_kernel.h).  Compiled once per (GCS_DIM, GCS_F32) pair as separate objects, in parallel: these
// instantiations take a minute or more each.
#include "vertex_kernel.h"

#ifndef GCS_DIM
#error "compile with -DGCS_DIM=3|6 and -DGCS_F32=0|1"
#endif

using namespace gcsadmm_k;

#if GCS_F32
typedef float state_t;
#else
typedef double state_t;
#endif

#define GCS_CAT2(a, b, c, d) a##b##c##d
#define GCS_CAT(a, b, c, d) GCS_CAT2(a, b, c, d)

void GCS_CAT(gcsadmm_launch_vertex_n, GCS_DIM, _f32_, GCS_F32)(const VertexLaunchDesc &d, hipStream_t s)
{
    launch_vertex_dim<GCS_DIM, state_t>(d, s);
}

hipError_t GCS_CAT(gcsadmm_lds_attr_n, GCS_DIM, _f32_, GCS_F32)(int lds_bytes)
{
    return set_lds_attr<GCS_DIM, state_t>(false, lds_bytes);
}
