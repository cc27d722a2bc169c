// gcs_math.h -- scalar helpers shared by the workgroup-cooperative vertex program (vertex_wg.h).
// On the device the f64 reciprocal / reciprocal square root are the hardware estimates refined by Newton steps
// (an IEEE division costs ~100 dependent cycles on gfx950, a refined estimate ~25; tools/micro/rcp_accuracy.hip);
// on the host (debug emulation, tests/hostemu) they are the plain IEEE operations.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GCS_HD __host__ __device__ __forceinline__
#else
#define GCS_HD inline
#endif

namespace gcs_math {

GCS_HD double rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// one Newton step (relative error ~2e-15): slack reciprocals and step-length ratios
GCS_HD double rcp1(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
#else
    return 1.0 / x;
#endif
}

GCS_HD double rsqrt_nr(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(x);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    return r;
#else
    return 1.0 / sqrt(x);
#endif
}

GCS_HD double sqrt_nr(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (!(x > 0.0)) return x < 0.0 ? __builtin_nan("") : x;
    const double r = rsqrt_nr(x);
    const double y = x * r;
    return fma(0.5 * r, fma(-y, y, x), y);
#else
    return sqrt(x);
#endif
}

// ---- second-order cone of dimension Q (first component = the epigraph variable) ----
template <int Q> GCS_HD double soc_det(const double *s)
{
    double nn = 0;
    for (int k = 1; k < Q; ++k) nn += s[k] * s[k];
    nn = sqrt_nr(nn);
    return (s[0] - nn) * (s[0] + nn);
}
template <int Q> GCS_HD bool soc_interior(const double *s)
{
    double nn = 0;
    for (int k = 1; k < Q; ++k) nn += s[k] * s[k];
    return s[0] > sqrt_nr(nn);
}
// largest step along ds that keeps s inside the cone (1e300: unbounded)
template <int Q> GCS_HD double soc_max_step(const double *s, const double *ds)
{
    double a = ds[0] * ds[0], b = s[0] * ds[0];
    const double c = soc_det<Q>(s);
    for (int k = 1; k < Q; ++k) { a -= ds[k] * ds[k]; b -= s[k] * ds[k]; }
    b *= 2;
    double al = 1e300;
    if (ds[0] < 0) al = fmin(al, -s[0] * rcp(ds[0]));
    if (fabs(a) < 1e-300) {
        if (b < 0) al = fmin(al, -c * rcp(b));
    } else {
        const double disc = b * b - 4 * a * c;
        if (disc >= 0) {
            const double sq = sqrt_nr(disc);
            const double qq = -0.5 * (b + (b >= 0 ? sq : -sq));
            const double r1 = qq * rcp(a), r2 = (qq != 0.0) ? c * rcp(qq) : 1e300;
            if (r1 > 0) al = fmin(al, r1);
            if (r2 > 0) al = fmin(al, r2);
        }
    }
    return al;
}

} // namespace gcs_math
