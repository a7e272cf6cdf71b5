#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *r, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i], y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0); e = fma(e, e, e); r[i] = fma(y, e, y);
}
int main() {
    const int n = 1 << 20; std::vector<double> x(n), c(n); std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-30, 30);
    for (auto &v : x) v = std::ldexp(1.0 + (g() >> 11) * 0x1p-53, (int)u(g)) * ((g() & 1) ? 1 : -1);
    double *dx, *d0; (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&d0, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); k<<<n / 256, 256>>>(dx, d0, n);
    (void)hipMemcpy(c.data(), d0, n * 8, hipMemcpyDeviceToHost);
    double m = 0; long ne = 0;
    for (int i = 0; i < n; ++i) { long double t = 1.0L / (long double)x[i]; m = std::fmax(m, (double)fabsl(((long double)c[i] - t) / t)); ne += c[i] != (double)t; }
    printf("cubic step: max rel err %.3g (%.2f ulp), != correctly rounded in %ld of %d\n", m, m / 0x1p-53, ne, n);
}
