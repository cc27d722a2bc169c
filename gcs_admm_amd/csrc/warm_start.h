// warm_start.h -- warm start of the per-vertex interior-point solves, shared by both vertex programs (vertex_wg.h,
// vertex_program.inc).  Same rule, same constants as oracle/gcs_oracle.c (WS_*), where it is derived and measured.
//
// The reference starts MOSEK cold every ADMM iteration (admm_solver_v3.py:490).  The minimiser does not depend on the start, and
// the feasible set of a vertex sub-problem depends neither on the targets nor on rho, so an interior iterate of the previous ADMM
// iteration's solve is a strictly feasible start of the next one:
//   record  = the first iterate, after at least one Newton step, whose barrier parameter is <= WS_SAVE * mu_ref: the primal point,
//             the equality multipliers and every dual, with rho and the targets of that solve, in the handle's HBM workspace;
//   restart = from the record when it is valid, rho is unchanged and dT = rho * max |T - T_record| <= theta_v (over the penalised
//             words), with mu_ref = max(WS_MU_MIN, WS_KAPPA * dT) (cold solves: WS_COLD_REF); the cone pair is re-centred in closed
//             form at mu_ref (t^2 - mu_ref t - |u|^2 = 0, lambda = (1, -u / t)); the FIRST iteration of a warm solve is a plain
//             Newton step towards s o lambda = mu_ref e (no predictor, no second-order term, no stop test), Mehrotra's iterations
//             follow as usual;
//   a warm solve that fails is repeated cold inside the same call;
//   theta_v = the vertex's own far-warm threshold, kept in its record beside n_cold, the iterations of its last cold solve (ws_theta,
//             ws_learn below): it starts at WS_COLD_DT and stays within [WS_NEAR, WS_THETA_MAX]; a solve that started cold because
//             the targets had moved too far raises it by WS_GROW; a warm solve from dT > WS_NEAR that took more iterations than
//             n_cold, or failed, lowers it to WS_SHRINK * dT.  (Whether a far record beats a cold start depends on the vertex --
//             benchmark4: 12.6 iterations against 9.5 for dT in [0.3, 1); lattices: 6.8 against 9.3 -- and a launch waits for its
//             slowest solve.)
// Record of one vertex with d incident edges and m facets, in doubles (wd_* below):
//   [0] valid  [1] rho  [2] theta_v  [3] n_cold   | x_v (2n) | nu (2 (2n+1)) | pad |      (t and the cone's dual are re-centred at the restart: not kept)
//   unit 0 .. d (unit 0 = border (z_v, y_v), unit e = block (O_e, y_e)):  p (2n+1) | bound duals (2) | targets (2n+1) | row duals (4m, as FLOATS)
//   row duals in the order  type (a: rows 1/3, b: rows 2/4) x half x facet.  They are kept in f32 (round 4): they are two thirds of a record and
//   a restart does not need more of them -- its first iteration re-centres the iterate at mu_ref >= 1e-7 anyway, and a relative 6e-8 on a dual
//   moves s o lambda by as much; the primal point, the multipliers of the equalities and the targets stay f64 (a slack can be ~1e-6: its
//   positivity must survive the round trip).  2.2 -> 1.6 KB per vertex on the n = 2 lattices, 5.8 -> 4.1 KB at n = 6.
#pragma once
#include <stdint.h>

namespace gcs_ws {

constexpr double WS_KAPPA = 3e-3, WS_MU_MIN = 1e-7, WS_COLD_DT = 1.0, WS_SAVE = 10.0, WS_COLD_REF = 1e-4;
constexpr double WS_NEAR = 0.1, WS_GROW = 1.25, WS_SHRINK = 0.5, WS_THETA_MAX = 10.0;

#if defined(__HIPCC__)
#define GCS_WS_HD __host__ __device__ __forceinline__
#else
#define GCS_WS_HD inline
#endif
// far-warm threshold of the record's vertex (a zeroed record: the default)
GCS_WS_HD double ws_theta(const double *rec)
{
    const double th = rec[2] > 0.0 ? rec[2] : WS_COLD_DT;
    return th > WS_NEAR ? th : WS_NEAR;
}
// after a solve that converged in `iters` iterations (of its last attempt).  warm: it restarted from the record; warm_failed: it did,
// failed, and was repeated cold; dT < 0: there was no comparable record (none yet, or rho changed)
GCS_WS_HD void ws_learn(double *rec, bool warm, bool warm_failed, double dT, int iters)
{
    if (!warm) {
        const double th = ws_theta(rec);
        rec[3] = (double)iters;
        if (warm_failed) rec[2] = WS_SHRINK * dT > WS_NEAR ? WS_SHRINK * dT : WS_NEAR;
        else if (dT >= 0.0) rec[2] = WS_GROW * th < WS_THETA_MAX ? WS_GROW * th : WS_THETA_MAX;
    } else if (dT > WS_NEAR && rec[3] > 0.0 && (double)iters > rec[3]) rec[2] = WS_SHRINK * dT > WS_NEAR ? WS_SHRINK * dT : WS_NEAR;
}
#undef GCS_WS_HD

constexpr int WD_HDR = 4;
constexpr int wd_pad2(int x) { return (x + 1) & ~1; }
template <int N> struct WRec {
    static constexpr int NW = 2 * N + 1, NX = 2 * N, Q = N + 1;
    static constexpr int XV = WD_HDR, NU = XV + NX, UNITS = wd_pad2(NU + 2 * NW);
    // within a unit
    static constexpr int P = 0, LB = NW, TG = NW + 2, LAM = 2 * NW + 2;      // LAM: 4m floats (= 2m doubles)
    static constexpr int unit_stride(int m) { return 2 * NW + 2 + 2 * m; }
    static constexpr long long doubles(int m, int d) { return UNITS + (long long)(d + 1) * unit_stride(m); }
};
inline long long warm_record_doubles(int n, int m, int d)
{
    const int NW = 2 * n + 1;
    return wd_pad2(WD_HDR + 2 * n + 2 * NW) + (long long)(d + 1) * (2 * NW + 2 + 2 * m);
}

}  // namespace gcs_ws
