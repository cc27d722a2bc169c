// Accuracy of the gfx950 f64 reciprocal / reciprocal-root estimates, raw and after Newton steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#pragma clang diagnostic ignored "-Wunused-result"
__global__ void acc(double *err, int n)
{
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = exp2(-40.0 + 80.0 * i / n) * (1.0 + 0.37 * (i % 977) / 977.0);
        const double ex = 1.0 / x, es = 1.0 / sqrt(x);
        double r = __builtin_amdgcn_rcp(x);
        e[0] = fmax(e[0], fabs(r - ex) / ex);
        r = fma(fma(-x, r, 1.0), r, r);
        e[1] = fmax(e[1], fabs(r - ex) / ex);
        r = fma(fma(-x, r, 1.0), r, r);
        e[2] = fmax(e[2], fabs(r - ex) / ex);
        double q = __builtin_amdgcn_rsq(x);
        e[3] = fmax(e[3], fabs(q - es) / es);
        q = fma(0.5 * q, fma(-x * q, q, 1.0), q);
        e[4] = fmax(e[4], fabs(q - es) / es);
        q = fma(0.5 * q, fma(-x * q, q, 1.0), q);
        e[5] = fmax(e[5], fabs(q - es) / es);
    }
    for (int k = 0; k < 6; ++k) err[(blockIdx.x * blockDim.x + threadIdx.x) * 6 + k] = e[k];
}
int main()
{
    const int T = 64 * 256;
    double *d; hipMalloc(&d, T * 6 * 8);
    acc<<<64, 256>>>(d, 1 << 24);
    static double h[T * 6]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double m[6] = {0};
    for (int i = 0; i < T; ++i) for (int k = 0; k < 6; ++k) m[k] = fmax(m[k], h[i * 6 + k]);
    printf("v_rcp_f64 max rel err: raw %.3e, 1 Newton %.3e, 2 Newton %.3e\n", m[0], m[1], m[2]);
    printf("v_rsq_f64 max rel err: raw %.3e, 1 Newton %.3e, 2 Newton %.3e\n", m[3], m[4], m[5]);
    return 0;
}
