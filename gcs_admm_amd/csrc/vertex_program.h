// vertex_program.h -- instantiates the wavefront program of the vertex step (vertex_program.inc) twice:
//   namespace gcs     generic facet count per polytope, facet-row duals in LDS
//   namespace gcs_m4  every polytope has exactly 4 facets (boxes in 2-D): facet loops fully unrolled, row
//                     duals in registers (8 KB less LDS per wavefront -> one more wavefront per CU)
#pragma once
#define GCS_NS gcs
#define GCS_MFIX 0
#include "vertex_program.inc"
#undef GCS_NS
#undef GCS_MFIX
#undef GCS_DUAL_REFS
#define GCS_NS gcs_m4
#define GCS_MFIX 4
#include "vertex_program.inc"
#undef GCS_NS
#undef GCS_MFIX
#undef GCS_DUAL_REFS
