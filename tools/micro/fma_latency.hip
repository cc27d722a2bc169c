// Dependent-issue latency of f64 FMA / LDS round trip / ds_bpermute at one wavefront per SIMD (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/fma_latency.hip -o tools/micro/fma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
template <int ILP> __global__ void fma_chain(double *out, long long *cyc, int iters, double a, double b)
{
    double x[ILP];
    for (int k = 0; k < ILP; ++k) x[k] = threadIdx.x + k;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) x[k] = fma(x[k], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int k = 0; k < ILP; ++k) s += x[k];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void lds_chain(double *out, long long *cyc, int iters)
{
    __shared__ double buf[64 * 8];
    for (int k = 0; k < 8; ++k) buf[threadIdx.x * 8 + k] = (double)((threadIdx.x * 8 + k + 1) % 512);
    __syncthreads();
    int idx = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) idx = (int)buf[idx];
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = idx;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void shfl_chain(double *out, long long *cyc, int iters)
{
    double x = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) x += __shfl_down(x, 1);
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void rsq_chain(double *out, long long *cyc, int iters)
{
    double x = 1.0 + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) x = __builtin_amdgcn_rsq(x) + 1.0;
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int ILP, int KIND> __global__ void trans_chain(double *out, long long *cyc, int iters)
{
    double x[ILP];
    for (int k = 0; k < ILP; ++k) x[k] = 1.5 + threadIdx.x + k;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            if (KIND == 0) x[k] = __builtin_amdgcn_rcp(x[k]);
            if (KIND == 1) x[k] = __builtin_amdgcn_rsq(x[k]);
            if (KIND == 2) x[k] = __builtin_amdgcn_sqrt(x[k]);
            if (KIND == 3) x[k] = (double)__builtin_amdgcn_rcpf((float)x[k]);
            if (KIND == 4) x[k] = 1.0 / x[k];
            if (KIND == 5) x[k] = sqrt(x[k]);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int k = 0; k < ILP; ++k) s += x[k];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NB> __global__ void lds_batch(double *out, long long *cyc, int iters)
{
    __shared__ double buf[1024];
    for (int k = threadIdx.x; k < 1024; k += 64) buf[k] = (double)((k * 7 + 1) % 1000);
    __syncthreads();
    int idx = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) s += buf[(idx + k * 17) & 1023];
        idx = (int)s & 1023;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = idx;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__device__ inline double dpp_shl1(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int NV, int KIND> __global__ void reduce_tp(double *out, long long *cyc, int iters)
{
    double x[NV];
    for (int k = 0; k < NV; ++k) x[k] = 1.5 + threadIdx.x + k;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (KIND == 0) x[k] += __shfl_down(x[k], 1);
            if (KIND == 1) x[k] += dpp_shl1(x[k]);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int k = 0; k < NV; ++k) s += x[k];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int it = 1000;
#define RUN(label, launch, per) launch; launch; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.2f cycles per op (s_memtime ticks)\n", label, (double)h / (per));
    RUN("fma f64 dependent", (fma_chain<1><<<1, 64>>>(out, cyc, it, 0.999, 0.5)), it * 16.0)
    RUN("fma f64 2 chains", (fma_chain<2><<<1, 64>>>(out, cyc, it, 0.999, 0.5)), it * 32.0)
    RUN("fma f64 4 chains", (fma_chain<4><<<1, 64>>>(out, cyc, it, 0.999, 0.5)), it * 64.0)
    RUN("fma f64 8 chains", (fma_chain<8><<<1, 64>>>(out, cyc, it, 0.999, 0.5)), it * 128.0)
    RUN("lds dependent read", (lds_chain<<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("shfl_down + add f64", (shfl_chain<<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("rsq f64 + add", (rsq_chain<<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("v_rcp_f64 dependent", (trans_chain<1, 0><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("v_rcp_f64 4 indep", (trans_chain<4, 0><<<1, 64>>>(out, cyc, it)), it * 4.0)
    RUN("v_rsq_f64 dependent", (trans_chain<1, 1><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("v_rsq_f64 4 indep", (trans_chain<4, 1><<<1, 64>>>(out, cyc, it)), it * 4.0)
    RUN("v_sqrt_f64 dependent", (trans_chain<1, 2><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("cvt+v_rcp_f32+cvt dependent", (trans_chain<1, 3><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("cvt+v_rcp_f32+cvt 4 indep", (trans_chain<4, 3><<<1, 64>>>(out, cyc, it)), it * 4.0)
    RUN("1.0/x IEEE dependent", (trans_chain<1, 4><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("sqrt IEEE dependent", (trans_chain<1, 5><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("lds 1 read + cvt chain", (lds_batch<1><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("lds 4 reads batch", (lds_batch<4><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("lds 16 reads batch", (lds_batch<16><<<1, 64>>>(out, cyc, it)), (double)it)
    RUN("shfl_down+add, 16 values", (reduce_tp<16, 0><<<1, 64>>>(out, cyc, it)), it * 16.0)
    RUN("dpp row_shl:1+add, 16 values", (reduce_tp<16, 1><<<1, 64>>>(out, cyc, it)), it * 16.0)
    RUN("dpp row_shl:1+add, 1 value", (reduce_tp<1, 1><<<1, 64>>>(out, cyc, it)), it * 1.0)
    // wall-clock calibration of the counter
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); fma_chain<1><<<1, 64>>>(out, cyc, 100000, 0.999, 0.5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("counter: %lld ticks in %.3f ms -> %.1f MHz; dependent fma = %.2f ns\n", h, ms, h / ms / 1e3, ms * 1e6 / (100000 * 16.0));
    return 0;
}
