// Cost of LDS instructions at one wavefront per SIMD (gfx950): N independent reads or writes with constant
// offsets from one per-lane base (the access pattern of the vertex kernel's slot arrays), issued back to back,
// one s_waitcnt at the end.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_ops.hip -o tools/micro/lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
template <int KIND> __global__ void k(double *out, long long *cyc, int iters, int active, int stride)
{
    extern __shared__ double buf[];
    for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = i;
    __syncthreads();
    double acc = 0;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        unsigned base = (unsigned)(threadIdx.x * stride * 8);
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) {        // 16 x ds_read_b64
                double v[16];
#define R(j) asm volatile("ds_read_b64 %0, %1 offset:" #j : "=v"(v[j / 8]) : "v"(base));
                R(0) R(8) R(16) R(24) R(32) R(40) R(48) R(56) R(64) R(72) R(80) R(88) R(96) R(104) R(112) R(120)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int j = 0; j < 16; ++j) acc += v[j];
            } else if (KIND == 1) { // 8 x ds_read_b128
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 v[8];
#define R2(j) asm volatile("ds_read_b128 %0, %1 offset:" #j : "=v"(v[j / 16]) : "v"(base));
                R2(0) R2(16) R2(32) R2(48) R2(64) R2(80) R2(96) R2(112)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
            } else if (KIND == 2) { // 16 x ds_write_b64
                double v = acc + i;
#define W(j) asm volatile("ds_write_b64 %0, %1 offset:" #j :: "v"(base), "v"(v) : "memory");
                W(0) W(8) W(16) W(24) W(32) W(40) W(48) W(56) W(64) W(72) W(80) W(88) W(96) W(104) W(112) W(120)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (KIND == 3) { // 8 x ds_write_b128
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 v = {acc + i, acc - i};
#define W2(j) asm volatile("ds_write_b128 %0, %1 offset:" #j :: "v"(base), "v"(v) : "memory");
                W2(0) W2(16) W2(32) W2(48) W2(64) W2(80) W2(96) W2(112)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (KIND == 4) { // 1 x ds_read_b64 + wait (latency)
                double v;
                asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base) : "memory");
                acc += v;
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int it = 2000;
    const char *names[5] = {"16 x ds_read_b64  (128 B/lane)", " 8 x ds_read_b128 (128 B/lane)", "16 x ds_write_b64 (128 B/lane)", " 8 x ds_write_b128 (128 B/lane)", " 1 x ds_read_b64 + wait"};
    for (int active : {64, 7, 1}) for (int stride : {16, 447}) {
        printf("active lanes %d, lane stride %d doubles\n", active, stride);
        for (int kind = 0; kind < 5; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) k<0><<<1, 64, 65536>>>(out, cyc, it, active, stride);
                if (kind == 1) k<1><<<1, 64, 65536>>>(out, cyc, it, active, stride);
                if (kind == 2) k<2><<<1, 64, 65536>>>(out, cyc, it, active, stride);
                if (kind == 3) k<3><<<1, 64, 65536>>>(out, cyc, it, active, stride);
                if (kind == 4) k<4><<<1, 64, 65536>>>(out, cyc, it, active, stride);
            }
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("  %-34s %8.1f cycles per batch\n", names[kind], (double)h / it);
        }
    }
    return 0;
}
