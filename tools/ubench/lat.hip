// Dependent-chain latency microbenchmarks for the kernel's building blocks on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 lat.hip -o lat ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
template <int K> __device__ __forceinline__ double bc(double v) {
    long long x = __double_as_longlong(v);
    x = __builtin_amdgcn_mov_dpp(x, 0x150 + K, 0xF, 0xF, false);
    return __longlong_as_double(x);
}
__device__ __forceinline__ double shf(double v, int src) {
    int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int MODE> __global__ void k(double *out, long long *cyc, double seed, int src) {
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    __shared__ double sh[256];
    sh[threadIdx.x] = a;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_ITER; ++i) {
        if (MODE == 0) { a = fma(a, b, c); }                               // dependent f64 FMA
        if (MODE == 1) { a = fma(a, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); }  // 4 independent
        if (MODE == 2) { a = fma(bc<3>(a), b, c); }                        // dpp -> fma dependent pair
        if (MODE == 3) { a = fma(shf(a, src), b, c); }                     // bpermute x2 -> fma
        if (MODE == 4) { double x = __builtin_amdgcn_rcp(a); a = fma(x, b, a); }  // rcp -> fma
        if (MODE == 5) { a = fma(sh[(threadIdx.x + (int)a) & 255], b, c); }       // LDS read (data-dependent addr) -> fma
        if (MODE == 6) { bool p = a > a2; unsigned long long m = __builtin_amdgcn_ballot_w64(p); if (m == 0x123456789ull) a2 += 1; a = fma(a, b, c); } // cmp+ballot+branch
        if (MODE == 7) { a = (threadIdx.x & 1) ? fma(a, b, c) : a; }        // fma + select
        if (MODE == 8) { a = exp(a * 1e-3) ; }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int ops, int waves_per_simd) {
    double *out; long long *cyc;
    int blocks = 256, threads = 256 * waves_per_simd;  // one block per CU
    if (threads > 1024) { blocks *= threads / 1024; threads = 1024; }
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    k<MODE><<<blocks, threads>>>(out, cyc, 1.0, 5);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // s_memtime counts at 100 MHz constant clock on gfx9? report raw and per-iteration
    printf("%-28s waves/SIMD %d: %.1f ticks/iter (%d op(s))\n", name, waves_per_simd, avg / N_ITER, ops);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("fma_f64 dependent", 1, w);
        run<1>("fma_f64 x4 independent", 4, w);
        run<2>("dpp64 -> fma", 2, w);
        run<3>("bpermute x2 -> fma", 3, w);
        run<4>("rcp_f64 -> fma", 2, w);
        run<5>("lds read -> fma", 2, w);
        run<6>("cmp+ballot+branch+fma", 4, w);
        run<7>("fma + select", 3, w);
        run<8>("exp(f64)", 1, w);
    }
    return 0;
}
