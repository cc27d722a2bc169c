// terminal_region.hip -- x-update of terminals that are regions (terminal_region.h): one 256-thread workgroup per terminal, launched on
// the handle's auxiliary stream beside the vertex-step launch (gcsadmm.hip launch_vertex).  Own object: the vertex kernels are not touched.
#include <hip/hip_runtime.h>

#include "gcsadmm.h"
#include "terminal_launch.h"
#include "terminal_region.h"

namespace gcsadmm_k {

constexpr int TERM_THREADS = 256;

// executor of terminal_region_solve on the device: the workgroup, reductions through LDS in a fixed order (bit-reproducible)
struct TermExec {
    double *red;      // [TERM_THREADS]
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return TERM_THREADS; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    template <class OP> __device__ __forceinline__ double reduce(double x, OP op)
    {
        red[threadIdx.x] = x;
        __syncthreads();
        for (int off = TERM_THREADS / 2; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] = op(red[threadIdx.x], red[threadIdx.x + off]);
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    }
    __device__ __forceinline__ double sum(double x) { return reduce(x, [](double a, double b) { return a + b; }); }
    __device__ __forceinline__ double min(double x) { return reduce(x, [](double a, double b) { return fmin(a, b); }); }
    __device__ __forceinline__ bool any(bool b) { return __syncthreads_or(b ? 1 : 0) != 0; }
};

template <int N, class T>
__global__ __launch_bounds__(TERM_THREADS) void terminal_region_kernel(TermLaunchDesc d)
{
    if (d.cb->status != GCSADMM_RUNNING) return;
    __shared__ gcs_term::TermShared<N> sh;
    __shared__ double red[TERM_THREADS];
    const int ti = (int)blockIdx.x, v = d.vtx[ti];
    gcs_term::TermProblem<T> P;
    const int p0 = d.poly_ptr[v], lo = d.inc_ptr[v];
    P.m = d.poly_ptr[v + 1] - p0; P.d = d.inc_ptr[v + 1] - lo; P.d_in = d.deg_in[v]; P.is_src = d.is_src[ti];
    P.A = d.poly_A + (size_t)p0 * N; P.bc = d.poly_bc + p0; P.cen = d.center + (size_t)v * N;
    P.inc_edge = d.inc_edge + lo; P.inc_lo = lo;
    P.E = d.E; P.NI = d.NI; P.edge_major = d.edge_major;
    P.zedge = (const T *)d.zedge; P.mu = (const T *)d.mu; P.copy = (T *)d.copy;
    P.xv = d.xv + (size_t)v * 2 * N; P.zv = d.zv + (size_t)v * 2 * N; P.yv = d.yv + v;
    P.rho = d.cb->rho; P.mu_scale = d.cb->mu_scale; P.eps_edge = d.eps_edge; P.ipm_tol = d.ipm_tol; P.ipm_max_iter = d.ipm_max_iter;
    TermExec ex{red};
    const int r = gcs_term::terminal_region_solve<N, T>(ex, P, d.ws + d.ws_off[ti], sh);
    if (threadIdx.x == 0) {
        if (r < 0) atomicAdd(&d.counters[0], 1);
        else atomicAdd(&d.counters[1], r);
    }
}

template <int N> static void launch_n(const TermLaunchDesc &d, hipStream_t s)
{
    if (d.dtype == GCSADMM_F64) hipLaunchKernelGGL((terminal_region_kernel<N, double>), dim3(d.count), dim3(TERM_THREADS), 0, s, d);
    else hipLaunchKernelGGL((terminal_region_kernel<N, float>), dim3(d.count), dim3(TERM_THREADS), 0, s, d);
}

}  // namespace gcsadmm_k

long long gcsadmm_terminal_ws_doubles(int n, int facets, int live_edges) { return gcs_term::terminal_ws_doubles(n, facets, live_edges); }

void gcsadmm_terminal_launch(const gcsadmm_k::TermLaunchDesc &d, hipStream_t s)
{
    using namespace gcsadmm_k;
    if (d.count <= 0) return;
    switch (d.n) {
    case 1: launch_n<1>(d, s); break;
    case 2: launch_n<2>(d, s); break;
    case 3: launch_n<3>(d, s); break;
    case 4: launch_n<4>(d, s); break;
    case 5: launch_n<5>(d, s); break;
    case 6: launch_n<6>(d, s); break;
    default: break;
    }
}
