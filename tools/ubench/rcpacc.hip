// accuracy of v_rcp_f64 and of 1 / 2 Newton refinements against the correctly rounded 1/x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i], y = __builtin_amdgcn_rcp(d);
    r0[i] = y;
    double e = fma(-d, y, 1.0); y = fma(y, e, y); r1[i] = y;
    e = fma(-d, y, 1.0); y = fma(y, e, y); r2[i] = y;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-30, 30);
    for (auto &v : x) v = std::ldexp(1.0 + (g() >> 11) * 0x1p-53, (int)u(g)) * ((g() & 1) ? 1 : -1);
    double *dx, *d0, *d1, *d2;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&d0, n * 8); (void)hipMalloc(&d1, n * 8); (void)hipMalloc(&d2, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    (void)hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0; long ne2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        m0 = std::fmax(m0, (double)fabsl(((long double)a[i] - t) / t));
        m1 = std::fmax(m1, (double)fabsl(((long double)b[i] - t) / t));
        m2 = std::fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
        ne2 += c[i] != (double)t;
    }
    printf("max rel err: rcp %.3g (2^%.1f)  +1 NR %.3g (%.2f ulp)  +2 NR %.3g (%.2f ulp), 2NR != correctly rounded in %ld of %d\n",
           m0, std::log2(m0), m1, m1 / 0x1p-53, m2, m2 / 0x1p-53, ne2, n);
    return 0;
}
