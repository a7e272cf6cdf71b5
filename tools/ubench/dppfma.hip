// v_fmac_f64_dpp (row_newbcast) on gfx950: throughput / latency against the mov_dpp + fma pair, and a
// check that the fused form computes  acc + bcast_K(src) * mul  -- including back-to-back use of a
// register written by the previous instruction, with and without the 2 wait states the ISA asks for
// between a VALU write and a DPP read.   hipcc --offload-arch=gfx950 -O3 dppfma.hip -o dppfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define N_ITER 4096
template <int K> __device__ __forceinline__ double bc(double v) {
    long long x = __double_as_longlong(v);
    x = __builtin_amdgcn_mov_dpp(x, 0x150 + K, 0xF, 0xF, false);
    return __longlong_as_double(x);
}
template <int K> __device__ __forceinline__ void fmac_bc(double &acc, double src, double mul) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
template <int K> __device__ __forceinline__ void fmac_bc_nop(double &acc, double src, double mul) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
template <int MODE> __global__ void k(double *out, long long *cyc, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = 1e-7, a2 = a + 1, a3 = a + 2, a4 = a + 3;
    double s1 = a * 3, s2 = a * 5, s3 = a * 7, s4 = a * 11;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_ITER; ++i) {
        if (MODE == 0) { a = fma(bc<3>(s1), b, a); a2 = fma(bc<4>(s2), b, a2); a3 = fma(bc<5>(s3), b, a3); a4 = fma(bc<6>(s4), b, a4); }
        if (MODE == 1) { fmac_bc<3>(a, s1, b); fmac_bc<4>(a2, s2, b); fmac_bc<5>(a3, s3, b); fmac_bc<6>(a4, s4, b); }
        if (MODE == 2) { a = fma(bc<3>(a), b, a); }        // dependent: reads what it just wrote
        if (MODE == 3) { fmac_bc<3>(a, a, b); }            // same, fused, NO wait states
        if (MODE == 4) { fmac_bc_nop<3>(a, a, b); }        // same, fused, s_nop 1
        if (MODE == 5) { a = fma(a, b, a); a2 = fma(a2, b, a2); a3 = fma(a3, b, a3); a4 = fma(a4, b, a4); }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> double run(const char *name, int ops, int waves_per_simd, std::vector<double> *res = nullptr) {
    double *out; long long *cyc;
    int blocks = 256, threads = 256 * waves_per_simd;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    k<MODE><<<blocks, threads>>>(out, cyc, 1.0);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    if (res) { res->resize(threads); hipMemcpy(res->data(), out, sizeof(double) * threads, hipMemcpyDeviceToHost); }
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("%-44s waves/SIMD %d: %7.1f ticks/iter (%d op(s))\n", name, waves_per_simd, avg / N_ITER, ops);
    hipFree(out); hipFree(cyc);
    return avg / N_ITER;
}
int main() {
    for (int w : {1, 2}) {
        run<5>("4 independent fma_f64", 4, w);
        run<0>("4 independent (mov_b64_dpp + fma_f64)", 8, w);
        run<1>("4 independent fmac_f64_dpp", 4, w);
        run<2>("dependent mov_dpp + fma", 2, w);
        run<3>("dependent fmac_f64_dpp, no nop", 1, w);
        run<4>("dependent fmac_f64_dpp, s_nop 1", 1, w);
    }
    std::vector<double> r0, r1, r2, r3, r4;
    run<0>("check", 8, 1, &r0); run<1>("check", 4, 1, &r1);
    run<2>("check", 2, 1, &r2); run<3>("check", 1, 1, &r3); run<4>("check", 1, 1, &r4);
    int bad01 = 0, bad23 = 0, bad24 = 0;
    for (size_t i = 0; i < r0.size(); ++i) {
        bad01 += r0[i] != r1[i]; bad23 += r2[i] != r3[i]; bad24 += r2[i] != r4[i];
    }
    printf("mismatches: independent fused vs pair %d | dependent fused(no nop) vs pair %d | dependent fused(nop) vs pair %d (of %zu lanes)\n",
           bad01, bad23, bad24, r0.size());
    printf("sample values: pair %.17g fused %.17g | dep pair %.17g dep fused %.17g dep fused nop %.17g\n", r0[5], r1[5], r2[5], r3[5], r4[5]);
    return 0;
}
