// terminal_region.hip -- x-update of terminals that are regions (terminal_region.h): one workgroup of 64 or 256 threads per terminal, launched on
// the handle's auxiliary stream beside the vertex-step launch (gcsadmm.hip launch_vertex).  Own object: the vertex kernels are not touched.
#include <hip/hip_runtime.h>

#include "gcsadmm.h"
#include "terminal_launch.h"
#include "terminal_region.h"

namespace gcsadmm_k {

constexpr int TERM_THREADS = 256;

// executor of terminal_region_solve on the device: one workgroup of 64 or 256 threads; reductions by wavefront shuffles in a fixed order
// (bit-reproducible), across wavefronts through LDS
#ifdef GCS_TERM_TIMING
__device__ unsigned long long g_term_cycles[32], g_term_visits[32];      // diagnostic build: ticks per phase of the solve, workgroup 0
#endif
struct TermExec {
    double *red;      // [3 * 4]
#ifdef GCS_TERM_TIMING
    unsigned long long last = 0;
    __device__ __forceinline__ void stamp(int k)
    {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0 && blockIdx.x == 0 && last) { atomicAdd(&g_term_cycles[k], t - last); atomicAdd(&g_term_visits[k], 1ull); }
        last = __builtin_amdgcn_s_memtime();
    }
#else
    __device__ __forceinline__ void stamp(int) {}
#endif
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ void reduce3(double &mn, double &s1, double &s2)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = fmin(mn, __shfl_xor(mn, off, 64));
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        const int nw = (int)blockDim.x >> 6;
        if (nw > 1) {
            const int w = (int)threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { red[w] = mn; red[4 + w] = s1; red[8 + w] = s2; }
            __syncthreads();
            mn = red[0]; s1 = red[4]; s2 = red[8];
            for (int k = 1; k < nw; ++k) { mn = fmin(mn, red[k]); s1 += red[4 + k]; s2 += red[8 + k]; }
            __syncthreads();
        }
    }
};

// LDSWS: the work arrays live in dynamic LDS (its own instantiation, so that the compiler addresses them as LDS rather than through flat pointers)
template <int N, class T, bool LDSWS>
__global__ __launch_bounds__(TERM_THREADS) void terminal_region_kernel(TermLaunchDesc d)
{
    extern __shared__ __attribute__((aligned(16))) double term_lds[];
    if (d.cb->status != GCSADMM_RUNNING) return;
    __shared__ gcs_term::TermShared<N> sh;
    __shared__ double red[12];
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
    P.warm = d.rec ? d.rec + d.rec_off[ti] : nullptr;
    TermExec ex{red};
    // work arrays: LDS when the launch was given room for the larger of the terminals (d.lds_doubles), the HBM workspace otherwise
    int r;
    if constexpr (LDSWS) r = gcs_term::terminal_region_solve<N, T>(ex, P, term_lds, sh);
    else r = gcs_term::terminal_region_solve<N, T>(ex, P, d.ws + d.ws_off[ti], sh);
    if (threadIdx.x == 0) {
        if (r < 0) atomicAdd(&d.counters[0], 1);
        else atomicAdd(&d.counters[1], r);
    }
}

template <int N> static void launch_n(const TermLaunchDesc &d, hipStream_t s)
{
    const size_t lds = (size_t)d.lds_doubles * sizeof(double);
    if (d.dtype == GCSADMM_F64) {
        if (lds) hipLaunchKernelGGL((terminal_region_kernel<N, double, true>), dim3(d.count), dim3(d.threads), lds, s, d);
        else hipLaunchKernelGGL((terminal_region_kernel<N, double, false>), dim3(d.count), dim3(d.threads), 0, s, d);
    } else {
        if (lds) hipLaunchKernelGGL((terminal_region_kernel<N, float, true>), dim3(d.count), dim3(d.threads), lds, s, d);
        else hipLaunchKernelGGL((terminal_region_kernel<N, float, false>), dim3(d.count), dim3(d.threads), 0, s, d);
    }
}

}  // namespace gcsadmm_k

#ifdef GCS_TERM_TIMING
extern "C" int gcsadmm_debug_term_cycles(unsigned long long *cycles32, unsigned long long *visits32)
{
    int e = (int)hipMemcpyFromSymbol(cycles32, HIP_SYMBOL(gcsadmm_k::g_term_cycles), 32 * sizeof(unsigned long long));
    if (e == 0) e = (int)hipMemcpyFromSymbol(visits32, HIP_SYMBOL(gcsadmm_k::g_term_visits), 32 * sizeof(unsigned long long));
    return e;
}
#endif

long long gcsadmm_terminal_ws_doubles(int n, int facets, int live_edges) { return gcs_term::terminal_ws_doubles(n, facets, live_edges); }
long long gcsadmm_terminal_record_doubles(int n, int facets, int live_edges) { return gcs_term::terminal_record_doubles(n, facets, live_edges); }

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
    case 7: launch_n<7>(d, s); break;
    case 8: launch_n<8>(d, s); break;
    default: break;
    }
}
