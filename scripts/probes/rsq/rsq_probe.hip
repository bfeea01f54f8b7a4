// How good is v_rsq_f64 + ONE Newton step?  (the 16 x 16 diagonal step of the ring solve runs 96 reciprocal square roots per pixel, each v_rsq_f64 + two steps = 8 instructions)
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/rsq/rsq_probe.hip -o /tmp/rsq_probe && /tmp/rsq_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, int n, double *e0, double *e1, double *e2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = x[i], h = 0.5 * v;
    double y0 = __builtin_amdgcn_rsq(v);
    double y1 = y0 * fma(-h * y0, y0, 1.5);
    double y2 = y1 * fma(-h * y1, y1, 1.5);
    double y3 = y2 * fma(-h * y2, y2, 1.5);
    // residual 1 - x y^2 evaluated with an fma on the exact product's high part: rel. error of y ~ residual / 2
    auto res = [&](double y) { const double p = v * y; return 0.5 * fabs(fma(p, y, -1.0)); };
    e0[i] = fabs(y0 - y3) / y3; e1[i] = fmax(fabs(y1 - y3) / y3, 0.0); e2[i] = fabs(y2 - y3) / y3;
    (void)res;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = ldexp(1.0 + u, (int)(s % 61) - 30); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, n, d0, d1, d2);
    std::vector<double> e0(n), e1(n), e2(n);
    hipMemcpy(e0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(e1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(e2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m0 = fmax(m0, e0[i]); m1 = fmax(m1, e1[i]); m2 = fmax(m2, e2[i]); }
    printf("max relative distance from the 3-step value over %d inputs in [2^-30, 2^31): seed %.3e (2^%.1f), one step %.3e (%.2f ulp), two steps %.3e (%.2f ulp)\n", n, m0, log2(m0), m1,
           m1 / 1.11e-16, m2, m2 / 1.11e-16);
    return 0;
}
