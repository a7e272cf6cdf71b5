// ldsread.hip -- cost of the run kernel's LDS read batches on gfx950 with every SIMD of every CU busy:
// 24 8-byte values per lane, the 16 lanes of a row at consecutive addresses, the 4 rows of a wave at the
// SAME addresses (shared model image) or at different ones (per-instance images); as 24 ds_read_b64, as
// 12 ds_read2_b64 (what the compiler merges neighbouring reads into) and as 12 ds_read_b128 (pairs
// stored contiguously).   hipcc --offload-arch=gfx950 -O3 ldsread.hip -o ldsread
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE, int SHARED> __global__ void k(double *out, double seed) {
    __shared__ double sh[9216];
    for (int i = threadIdx.x; i < 9216; i += blockDim.x) sh[i] = seed + i;
    __syncthreads();
    const int lane = threadIdx.x & 63, lig = lane & 15, grp = lane >> 4;
    double acc = 0;
    for (int it = 0; it < N_ITER; ++it) {
        const int rot = (it & 3) * 2;
        if (MODE == 0) {
            const unsigned p = (unsigned)(size_t)(sh + (SHARED ? 0 : grp * 1024) + lig + rot);
            double v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, v16, v17, v18, v19, v20, v21, v22, v23;
            asm volatile("ds_read_b64 %0, %24 offset:0\n\tds_read_b64 %1, %24 offset:128\n\tds_read_b64 %2, %24 offset:256\n\tds_read_b64 %3, %24 offset:384\n\tds_read_b64 %4, %24 offset:512\n\tds_read_b64 %5, %24 offset:640\n\tds_read_b64 %6, %24 offset:768\n\tds_read_b64 %7, %24 offset:896\n\tds_read_b64 %8, %24 offset:1024\n\tds_read_b64 %9, %24 offset:1152\n\tds_read_b64 %10, %24 offset:1280\n\tds_read_b64 %11, %24 offset:1408\n\tds_read_b64 %12, %24 offset:1536\n\tds_read_b64 %13, %24 offset:1664\n\tds_read_b64 %14, %24 offset:1792\n\tds_read_b64 %15, %24 offset:1920\n\tds_read_b64 %16, %24 offset:2048\n\tds_read_b64 %17, %24 offset:2176\n\tds_read_b64 %18, %24 offset:2304\n\tds_read_b64 %19, %24 offset:2432\n\tds_read_b64 %20, %24 offset:2560\n\tds_read_b64 %21, %24 offset:2688\n\tds_read_b64 %22, %24 offset:2816\n\tds_read_b64 %23, %24 offset:2944\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7), "=&v"(v8), "=&v"(v9), "=&v"(v10), "=&v"(v11), "=&v"(v12), "=&v"(v13), "=&v"(v14), "=&v"(v15), "=&v"(v16), "=&v"(v17), "=&v"(v18), "=&v"(v19), "=&v"(v20), "=&v"(v21), "=&v"(v22), "=&v"(v23) : "v"(p));
            acc += v0 + v23;
        }
        if (MODE == 1) {
            const unsigned p = (unsigned)(size_t)(sh + (SHARED ? 0 : grp * 1024) + lig + rot);
            d2 q0, q1, q2, q3, q4, q5, q6, q7, q8, q9, q10, q11;
            asm volatile("ds_read2_b64 %0, %12 offset0:0 offset1:16\n\tds_read2_b64 %1, %12 offset0:32 offset1:48\n\tds_read2_b64 %2, %12 offset0:64 offset1:80\n\tds_read2_b64 %3, %12 offset0:96 offset1:112\n\tds_read2_b64 %4, %12 offset0:128 offset1:144\n\tds_read2_b64 %5, %12 offset0:160 offset1:176\n\tds_read2_b64 %6, %12 offset0:192 offset1:208\n\tds_read2_b64 %7, %12 offset0:224 offset1:240\n\tds_read2_b64 %8, %13 offset0:0 offset1:16\n\tds_read2_b64 %9, %13 offset0:32 offset1:48\n\tds_read2_b64 %10, %13 offset0:64 offset1:80\n\tds_read2_b64 %11, %13 offset0:96 offset1:112\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7), "=&v"(q8), "=&v"(q9), "=&v"(q10), "=&v"(q11) : "v"(p), "v"(p + 2048));
            acc += q0.x + q11.y;
        }
        if (MODE == 2) {   // pairs contiguous: lane lig reads 16 bytes at (pair*32 + lig*2)
            const unsigned p = (unsigned)(size_t)(sh + (SHARED ? 0 : grp * 1024) + 2 * lig + rot);
            d2 q0, q1, q2, q3, q4, q5, q6, q7, q8, q9, q10, q11;
            asm volatile("ds_read_b128 %0, %12 offset:0\n\tds_read_b128 %1, %12 offset:256\n\tds_read_b128 %2, %12 offset:512\n\tds_read_b128 %3, %12 offset:768\n\tds_read_b128 %4, %12 offset:1024\n\tds_read_b128 %5, %12 offset:1280\n\tds_read_b128 %6, %12 offset:1536\n\tds_read_b128 %7, %12 offset:1792\n\tds_read_b128 %8, %12 offset:2048\n\tds_read_b128 %9, %12 offset:2304\n\tds_read_b128 %10, %12 offset:2560\n\tds_read_b128 %11, %12 offset:2816\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7), "=&v"(q8), "=&v"(q9), "=&v"(q10), "=&v"(q11) : "v"(p));
            acc += q0.x + q11.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE, int SHARED> void run(const char *name, int waves_per_simd) {
    double *out;
    int blocks = 256, threads = 256 * waves_per_simd;
    (void)hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, SHARED><<<blocks, threads>>>(out, 1.0);
    (void)hipEventRecord(e0);
    k<MODE, SHARED><<<blocks, threads>>>(out, 1.0);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD %d: %7.1f ns per batch of 24 doubles per wave (%5.2f ns per double per SIMD)\n", name, waves_per_simd,
           ms * 1e6 / N_ITER, ms * 1e6 / N_ITER / 24 / waves_per_simd);
    (void)hipFree(out);
}
int main() {
    for (int w : {1, 2}) {
        run<0, 1>("24 x ds_read_b64, rows share addresses", w);
        run<0, 0>("24 x ds_read_b64, rows at different images", w);
        run<1, 1>("12 x ds_read2_b64, rows share addresses", w);
        run<1, 0>("12 x ds_read2_b64, rows at different images", w);
        run<2, 1>("12 x ds_read_b128, rows share addresses", w);
        run<2, 0>("12 x ds_read_b128, rows at different images", w);
    }
    return 0;
}
