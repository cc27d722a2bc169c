// vertex_program.h -- instantiates the wavefront program of the vertex step (vertex_program.inc) twice:
//   namespace gcs     generic facet count per polytope, facet-row duals in LDS
//   namespace gcs_m4  every polytope has exactly 4 facets: facet loops fully unrolled, row
//                     duals in registers (8 KB less LDS per wavefront -> one more wavefront per CU)
//   namespace gcs_box every polytope is an axis-aligned box in 2-D with canonical facet order [+e0, +e1, -e0, -e1] (the lattice
//                     configurations): as gcs_m4, and the facet normals are compile-time constants (no products with zeros)
#pragma once
#define GCS_NS gcs
#define GCS_MFIX 0
#define GCS_BOX 0
#include "vertex_program.inc"
#undef GCS_NS
#undef GCS_MFIX
#undef GCS_BOX
#undef GCS_DUAL_REFS
#define GCS_NS gcs_m4
#define GCS_MFIX 4
#define GCS_BOX 0
#include "vertex_program.inc"
#undef GCS_NS
#undef GCS_MFIX
#undef GCS_BOX
#undef GCS_DUAL_REFS
#define GCS_NS gcs_box
#define GCS_MFIX 4
#define GCS_BOX 1
#include "vertex_program.inc"
#undef GCS_NS
#undef GCS_MFIX
#undef GCS_BOX
#undef GCS_DUAL_REFS
