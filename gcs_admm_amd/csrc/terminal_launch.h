// terminal_launch.h -- host-side interface of the region-terminal kernel (terminal_region.hip), used by gcsadmm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "gcsadmm.h"

namespace gcsadmm_k {

struct TermLaunchDesc {
    int n, dtype;                   // space dimension 1 .. 6, GCSADMM_F64 / GCSADMM_F32
    int count;                      // region terminals of the handle: 0, 1 or 2 -- one workgroup each
    int vtx[2], is_src[2];
    long long ws_off[2];            // workspace of each terminal, ws + ws_off[i], gcsadmm_terminal_ws_doubles(n, facets, live edges) doubles
    double *ws;
    double *rec;                    // warm-start records (terminal_region.h), rec + rec_off[i]; nullptr: every solve starts cold
    long long rec_off[2];
    int threads;                    // 64 (one wavefront: small terminals) or 256
    int lds_doubles;                // > 0: the work arrays of every terminal fit this much dynamic LDS and live there; 0: in ws
    const int *inc_ptr, *deg_in, *inc_edge, *poly_ptr;
    const double *poly_A, *poly_bc, *center;
    int E, NI, edge_major;
    void *zedge, *mu, *copy;
    double *xv, *zv, *yv;
    int *counters;
    const gcsadmm_control_block *cb;
    double eps_edge, ipm_tol;
    int ipm_max_iter;
};

}  // namespace gcsadmm_k

long long gcsadmm_terminal_ws_doubles(int n, int facets, int live_edges);
long long gcsadmm_terminal_record_doubles(int n, int facets, int live_edges);
void gcsadmm_terminal_launch(const gcsadmm_k::TermLaunchDesc &d, hipStream_t s);
