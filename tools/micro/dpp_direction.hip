// Which way do the DPP row shifts move data on gfx950?  out[i] = value read by lane i.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
template <int CTRL> __global__ void k(int *out)
{
    const int x = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __global__ void k2(int *out)   // bound_ctrl = false: keeps `old` where the source lane is outside the row
{
    const int x = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, CTRL, 0xf, 0xf, false);
}
int main()
{
    int *d, h[64];
    hipMalloc(&d, 256);
    auto show = [&](const char *name) {
        hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("%-22s", name);
        for (int i = 0; i < 20; ++i) printf(" %4d", h[i]);
        printf(" ... lane 31: %d lane 32: %d lane 63: %d\n", h[31], h[32], h[63]);
    };
    k<0x101><<<1, 64>>>(d); show("row_shl:1 bc=1");
    k<0x104><<<1, 64>>>(d); show("row_shl:4 bc=1");
    k<0x111><<<1, 64>>>(d); show("row_shr:1 bc=1");
    k2<0x101><<<1, 64>>>(d); show("row_shl:1 bc=0");
    k<0x130><<<1, 64>>>(d); show("wave_shl:1 bc=1");
    return 0;
}
