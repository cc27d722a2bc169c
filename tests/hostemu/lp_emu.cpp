// Host build of the per-lane LP solver of csrc/polytope_lp.hip (polytope_lp_core.h), for CPU tests of the
// algorithm on a machine without a GPU.  Test infrastructure: never shipped, never timed.
#include <vector>
#include "polytope_lp_core.h"
using namespace gcsadmm_lp;

template <int N> static int centre(const Polys &S, int p, int q, const double *x0, double *w_out, int early, double tol, int *iters)
{
    Rows<N, true> R(S, p, q);
    double w[N + 1], c[N + 1];
    for (int k = 0; k < N; ++k) c[k] = 0.0;
    c[N] = -1.0;
    ball_start<N>(R, x0, w);
    std::vector<double> lam((size_t)R.m * WAVE), dlam((size_t)R.m * WAVE);
    const int st = lp_ipm<N, true>(R, c, w, lam.data(), dlam.data(), 0, early != 0, tol, iters);
    for (int k = 0; k <= N; ++k) w_out[k] = w[k];
    return st;
}

// ball LP over the rows of polytope p (and q if q >= 0); returns the LP status, w_out = (x, r)
extern "C" int lp_emu_ball(int n, int P, const int *ptr, const double *A, const double *b, const double *nrm, int p, int q,
                           const double *x0, int early, double tol, double *w_out, int *iters)
{
    Polys S{n, P, ptr, A, b, nrm};
    switch (n) {
    case 1: return centre<1>(S, p, q, x0, w_out, early, tol, iters);
    case 2: return centre<2>(S, p, q, x0, w_out, early, tol, iters);
    case 3: return centre<3>(S, p, q, x0, w_out, early, tol, iters);
    case 4: return centre<4>(S, p, q, x0, w_out, early, tol, iters);
    case 5: return centre<5>(S, p, q, x0, w_out, early, tol, iters);
    case 6: return centre<6>(S, p, q, x0, w_out, early, tol, iters);
    case 7: return centre<7>(S, p, q, x0, w_out, early, tol, iters);
    case 8: return centre<8>(S, p, q, x0, w_out, early, tol, iters);
    }
    return -99;
}
