// issue.hip -- issue cost (SIMD cycles per wave-instruction) of the instruction kinds the run kernel is
// made of, on gfx950: streams of INDEPENDENT instructions (throughput) and DEPENDENT chains (latency),
// with 1 and 2 waves per SIMD.  The kernel is bound by per-wave instruction issue: these
// numbers are its cost model.   Build: hipcc --offload-arch=gfx950 -O3 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define N_ITER 2048
// 8 independent accumulator pairs v[A..], operands b, c
template <int MODE> __global__ void k(double *out, long long *cyc, double seed) {
    double a0 = seed + threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001, c = 1e-9;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7, ib = 77;
    unsigned long long s0 = 0x0001000100010001ull, s1 = 0;
    __shared__ double sh[1024];
    sh[threadIdx.x] = a0;
    __syncthreads();
    const double *lp = sh + (threadIdx.x & 63);
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_ITER; ++i) {
#define OPS8(op) asm volatile(op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7) \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(s0));
#define DEP8(op) asm volatile(op(%0) op(%0) op(%0) op(%0) op(%0) op(%0) op(%0) op(%0) \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(s0));
#define FMA(r) "v_fma_f64 " #r ", " #r ", %8, %9\n\t"
#define FMAC(r) "v_fmac_f64_e32 " #r ", %8, %9\n\t"
#define FMACDPP(r) "v_fmac_f64_dpp " #r ", %9, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define FMACDPPSELF(r) "s_nop 1\n\tv_fmac_f64_dpp " #r ", " #r ", %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define MOVDPP(r) "v_mov_b64_dpp " #r ", %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define MOV64(r) "v_mov_b64 " #r ", %9\n\t"
#define MUL(r) "v_mul_f64 " #r ", " #r ", %8\n\t"
#define ADD(r) "v_add_f64 " #r ", " #r ", %9\n\t"
#define MAXF(r) "v_max_f64 " #r ", " #r ", %9\n\t"
#define RCP(r) "v_rcp_f64_e32 " #r ", " #r "\n\t"
#define CND(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, %10\n\t"
#define CMP(r) "v_cmp_gt_f64_e64 s[20:21], |" #r "|, 4.0\n\t"
#define SAND(r) "s_and_b32 s20, s20, 0x1ffe1ffe\n\t"
#define SNOP(r) "s_nop 0\n\t"
#define SNOP1(r) "s_nop 1\n\t"
#define EXECMOV(r) "s_and_saveexec_b64 s[20:21], %10\n\tv_mov_b64 " #r ", %9\n\ts_mov_b64 exec, s[20:21]\n\t"
#define LDEXP(r) "v_ldexp_f64 " #r ", " #r ", 1\n\t"
#define RNDNE(r) "v_rndne_f64_e32 " #r ", " #r "\n\t"
        if (MODE == 0) { REP4(OPS8(FMA)) }
        if (MODE == 1) { REP4(DEP8(FMA)) }
        if (MODE == 2) { REP4(OPS8(FMACDPP)) }
        if (MODE == 3) { REP4(OPS8(MOVDPP)) }
        if (MODE == 4) { REP4(OPS8(MOV64)) }
        if (MODE == 5) { REP4(OPS8(MUL)) }
        if (MODE == 6) { REP4(OPS8(RCP)) }
        if (MODE == 7) {
            REP4(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n\tv_cndmask_b32_e64 %1, %1, %8, %9\n\tv_cndmask_b32_e64 %2, %2, %8, %9\n\tv_cndmask_b32_e64 %3, %3, %8, %9\n\t"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n\tv_cndmask_b32_e64 %5, %5, %8, %9\n\tv_cndmask_b32_e64 %6, %6, %8, %9\n\tv_cndmask_b32_e64 %7, %7, %8, %9"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib), "s"(s0));)
        }
        if (MODE == 8) { REP4(OPS8(CMP)) }
        if (MODE == 9) { REP4(OPS8(SAND)) }
        if (MODE == 10) { REP4(OPS8(SNOP)) }
        if (MODE == 11) { REP4(OPS8(FMACDPPSELF)) }
        if (MODE == 12) { REP4(OPS8(EXECMOV)) }
        if (MODE == 13) { REP4(OPS8(ADD)) }
        if (MODE == 14) { REP4(OPS8(LDEXP)) }
        if (MODE == 15) { REP4(OPS8(RNDNE)) }
        if (MODE == 16) { REP4(DEP8(FMAC)) }
        if (MODE == 17) { REP4(OPS8(SNOP1)) }
        if (MODE == 18) {   // 32 LDS reads (conflict-free b64), waited for once
            double t0_, t1_, t2_, t3_, t4_, t5_, t6_, t7_;
            REP4(asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                         "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_), "=&v"(t4_), "=&v"(t5_), "=&v"(t6_), "=&v"(t7_) : "v"((unsigned)(size_t)lp));)
            a0 += t0_ + t7_;
        }
        if (MODE == 19) {   // 16 x ds_read2_b64
            double __attribute__((ext_vector_type(2))) q0, q1, q2, q3;
            REP4(asm volatile("ds_read2_b64 %0, %4 offset1:64\n\tds_read2_b64 %1, %4 offset0:128 offset1:192\n\tds_read2_b64 %2, %4 offset0:32 offset1:96\n\tds_read2_b64 %3, %4 offset0:160 offset1:224\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"((unsigned)(size_t)lp));)
            a0 += q0.x + q3.y;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)s1 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int ops_per_iter, int waves_per_simd) {
    double *out; long long *cyc;
    int blocks = 256, threads = 256 * waves_per_simd;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, cyc, 1.0);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, cyc, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // per SIMD: waves_per_simd waves each issue ops_per_iter*N_ITER instructions
    double ns_per_instr_per_simd = ms * 1e6 / ((double)ops_per_iter * N_ITER * waves_per_simd);
    printf("%-34s waves/SIMD %d: %7.2f counter ticks per instr per wave, %6.2f ns per instr per SIMD (event timing)\n", name, waves_per_simd,
           avg / N_ITER / ops_per_iter, ns_per_instr_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2}) {
        run<0>("v_fma_f64 independent", 32, w);
        run<1>("v_fma_f64 dependent chain", 32, w);
        run<16>("v_fmac_f64 dependent chain", 32, w);
        run<2>("v_fmac_f64_dpp independent", 32, w);
        run<11>("s_nop1 + v_fmac_f64_dpp self", 64, w);
        run<3>("v_mov_b64_dpp", 32, w);
        run<4>("v_mov_b64", 32, w);
        run<5>("v_mul_f64", 32, w);
        run<13>("v_add_f64", 32, w);
        run<6>("v_rcp_f64", 32, w);
        run<14>("v_ldexp_f64", 32, w);
        run<15>("v_rndne_f64", 32, w);
        run<7>("v_cndmask_b32 (sgpr mask)", 32, w);
        run<8>("v_cmp_gt_f64 -> sgpr", 32, w);
        run<9>("s_and_b32", 32, w);
        run<10>("s_nop 0", 32, w);
        run<17>("s_nop 1", 32, w);
        run<12>("saveexec + v_mov_b64 + restore", 96, w);
        run<18>("ds_read_b64 x8 + wait", 32, w);
        run<19>("ds_read2_b64 x4 + wait", 16, w);
    }
    return 0;
}
