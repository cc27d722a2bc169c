// term_emu.cpp -- the region-terminal solve of the HIP library (gcs_admm_amd/csrc/terminal_region.h) compiled for the host: the same body
// with a one-thread executor.  Test-only (tests/test_terminal_region.py); the GPU parity tests are in tests/test_gpu_configs.py.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "terminal_region.h"

namespace {
struct HostExec {
    int tid() const { return 0; }
    int nthreads() const { return 1; }
    void sync() {}
    void reduce3(double &, double &, double &) {}
    void stamp(int) {}
};

template <int N>
int run(int m, const double *A, const double *b_raw, const double *cen, int d, int d_in, int is_src, const double *T, double rho,
        double eps_edge, double ipm_tol, int ipm_max_iter, double *copy, double *xv, double *zv, double *yv, double *warm)
{
    std::vector<double> bc(m), zero((size_t)(2 * N + 1) * d, 0.0);
    std::vector<int> edges(d);
    for (int j = 0; j < m; ++j) { double a = b_raw[j]; for (int k = 0; k < N; ++k) a -= A[j * N + k] * cen[k]; bc[j] = a; }
    for (int e = 0; e < d; ++e) edges[e] = e;
    gcs_term::TermProblem<double> P;
    P.m = m; P.d = d; P.d_in = d_in; P.is_src = is_src; P.A = A; P.bc = bc.data(); P.cen = cen;
    P.inc_edge = edges.data(); P.inc_lo = 0; P.E = d; P.NI = d; P.edge_major = 0;
    P.zedge = T; P.mu = zero.data(); P.copy = copy; P.xv = xv; P.zv = zv; P.yv = yv;
    P.rho = rho; P.mu_scale = 0.0; P.eps_edge = eps_edge; P.ipm_tol = ipm_tol; P.ipm_max_iter = ipm_max_iter; P.warm = warm;
    const int L = is_src ? d - d_in : d_in;
    std::vector<double> ws((size_t)gcs_term::terminal_ws_doubles(N, m, L > 0 ? L : 1), NAN);      // unwritten workspace is poisoned
    gcs_term::TermShared<N> sh;
    HostExec ex;
    return gcs_term::terminal_region_solve<N, double>(ex, P, ws.data(), sh);
}
}  // namespace

extern "C" int term_emu_solve(int n, int m, const double *A, const double *b_raw, const double *cen, int d, int d_in, int is_src,
                              const double *T, double rho, double eps_edge, double ipm_tol, int ipm_max_iter, double *copy, double *xv,
                              double *zv, double *yv, double *warm)
{
    switch (n) {
    case 1: return run<1>(m, A, b_raw, cen, d, d_in, is_src, T, rho, eps_edge, ipm_tol, ipm_max_iter, copy, xv, zv, yv, warm);
    case 2: return run<2>(m, A, b_raw, cen, d, d_in, is_src, T, rho, eps_edge, ipm_tol, ipm_max_iter, copy, xv, zv, yv, warm);
    case 3: return run<3>(m, A, b_raw, cen, d, d_in, is_src, T, rho, eps_edge, ipm_tol, ipm_max_iter, copy, xv, zv, yv, warm);
    case 6: return run<6>(m, A, b_raw, cen, d, d_in, is_src, T, rho, eps_edge, ipm_tol, ipm_max_iter, copy, xv, zv, yv, warm);
    case 8: return run<8>(m, A, b_raw, cen, d, d_in, is_src, T, rho, eps_edge, ipm_tol, ipm_max_iter, copy, xv, zv, yv, warm);
    default: return -100;
    }
}

// doubles of a terminal's warm-start record (zeroed by the caller: no record yet)
extern "C" long long term_emu_record_doubles(int n, int m, int live) { return gcs_term::terminal_record_doubles(n, m, live); }
