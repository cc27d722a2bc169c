// canonical_box.h -- the facet pattern the box instantiations of the two vertex programs assume (host-side check at create)
#pragma once
namespace gcsadmm_k {
// true when the m x n facet matrix is exactly [I; -I]: facets in the order +e_0 .. +e_{n-1}, -e_0 .. -e_{n-1}
// (gcs_admm_amd.graph.lattice_boxes builds its boxes this way; the reference's convert_pt_to_polytope, utils.py:12-28, too)
inline bool canonical_box(int n, int m, const double *A)
{
    if (m != 2 * n) return false;
    for (int j = 0; j < m; ++j)
        for (int k = 0; k < n; ++k)
            if (A[j * n + k] != ((j % n) == k ? (j < n ? 1.0 : -1.0) : 0.0)) return false;
    return true;
}
}  // namespace gcsadmm_k
