// gcsadmm_dims.hip -- the vertex-step kernels for space dimensions 3 and 6 (same wavefront program as n = 2,
// see vertex_kernel.h).  Separate translation unit: these instantiations take minutes to compile.
#include "vertex_kernel.h"

using namespace gcsadmm_k;

void gcsadmm_launch_vertex_hi(int n, int dtype, const VertexLaunchDesc &d, hipStream_t s)
{
    if (n == 3) { if (dtype == GCSADMM_F64) launch_vertex_dim<3, double>(d, s); else launch_vertex_dim<3, float>(d, s); }
    else        { if (dtype == GCSADMM_F64) launch_vertex_dim<6, double>(d, s); else launch_vertex_dim<6, float>(d, s); }
}

hipError_t gcsadmm_lds_attr_hi(int n, int dtype, int lds_bytes)
{
    if (n == 3) return dtype == GCSADMM_F64 ? set_lds_attr<3, double>(false, lds_bytes) : set_lds_attr<3, float>(false, lds_bytes);
    return dtype == GCSADMM_F64 ? set_lds_attr<6, double>(false, lds_bytes) : set_lds_attr<6, float>(false, lds_bytes);
}
