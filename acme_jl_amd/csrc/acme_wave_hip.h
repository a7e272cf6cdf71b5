// acme_wave_hip.h -- gfx950 (CDNA4, wave64) cross-lane primitives used by the kernels.
//
// One circuit instance lives in one DPP row (16 lanes).  Row-local broadcasts and
// rotations are DPP modifiers on v_mov_b32 (row_newbcast / row_ror): no LDS traffic, no
// address VGPRs.  Only the dynamic row interchange of the pivoting LU needs ds_bpermute.
#pragma once
#include <hip/hip_runtime.h>

#define ACME_DEV __device__ __forceinline__

namespace wv {

// the 64 lanes of a wave execute every instruction together (the emulator's run one after the other between rendezvous)
// (-DACME_COOP_NO_MIRROR: a developer switch of the mid-size kernel, A/B of its mirror rows against rows that leave)
#ifdef ACME_COOP_NO_MIRROR
ACME_DEV constexpr bool lockstep() { return false; }
#else
ACME_DEV constexpr bool lockstep() { return true; }
#endif
ACME_DEV int tid() { return (int)threadIdx.x; }
ACME_DEV int bid() { return (int)blockIdx.x; }
ACME_DEV void block_sync() { __syncthreads(); }
// orders this wave's LDS writes before its later LDS reads (one wave executes DS ops in
// order; this only stops the compiler from reordering across it)
ACME_DEV void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                             __builtin_amdgcn_wave_barrier(); }
// the same ordering WITHOUT the s_waitcnt the release fence brings along: the DS pipeline serves a wave's operations in
// program order, so a read issued right behind a write of another lane of the SAME wave sees it -- where the hand-off's
// latency is the cost (a lone wave's pivot-row / x_j broadcasts through LDS), this saves one LDS round trip per hand-off
ACME_DEV void lds_order() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

// DPP moves with an undefined `old` operand: every lane of a row has a valid source for
// row_newbcast / row_ror, so no destination pre-initialisation (no extra v_mov) is needed.
template <int CTRL> ACME_DEV int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, false); }
// value of lane K of each 16-lane row, in every lane of that row.  row_newbcast is the one
// DPP control gfx950 accepts on 64-bit moves: ONE v_mov_b64_dpp per broadcast double.
template <int K> ACME_DEV double bcast16(double v) {
    long long x = __double_as_longlong(v);
    x = __builtin_amdgcn_mov_dpp(x, 0x150 + K, 0xF, 0xF, false);
    return __longlong_as_double(x);
}
template <int K> ACME_DEV int bcast16(int v) { return dpp_i<0x150 + K>(v); }

// Fused broadcast-multiply-add  acc += (lane K of acc's row) * mul  as ONE v_fmac_f64_dpp
// (8 bytes, one issue slot) instead of v_mov_b64_dpp + v_fmac_f64 (12 bytes, two slots): per-wave
// instruction fetch and issue, not the FP64 pipe, bound the elimination loops.
// A DPP read needs 2 wait states after a VALU write of its source.  The compiler's hazard
// recogniser does not look inside inline asm, so the order is made explicit: these statements
// are volatile (kept in program order), callers arrange that at least two of them -- or an
// unrelated dependent chain -- sit between a write and the DPP read of the same register, and
// use the SAFE forms (leading s_nop 1) wherever the producer may be compiler-scheduled code.
template <int K, bool SAFE> ACME_DEV void fmac_bcast_self(double &acc, double mul) {
    if (SAFE)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(mul), "n"(K));
    else
        asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(mul), "n"(K));
}
// acc += (lane K of src's row) * mul  with src another register than acc: the broadcast of x_j /
// z_j / p_j fused into the multiply-add that consumes it (one instruction instead of v_mov_b64_dpp +
// v_fmac_f64, and no register for the broadcast value).  src must not have been written by a VALU
// instruction within the two preceding wait states: callers pass values produced well before, and
// tools/dpp_hazard_check.py proves it for every DPP instruction of the built code object.
template <int K> ACME_DEV void fmac_bcast(double &acc, double src, double mul) {
    // NOT volatile: the compiler may interleave these with the LDS reads that feed them (as ordered
    // statements they made the wave wait for whole batches of loads: -4 % instead of +1.7 %)
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
// A whole chain  acc += sum_j (lane j of src's row) * mul[j],  j = 0 .. N-1,  as ONE asm statement.  The
// compiler assumes that a value written by inline asm has the dst-forwarding hazard (gfx940+) and puts an
// s_nop between any asm statement and a consumer of its result: with one statement per multiply-add that
// is a wait state after EVERY v_fmac_f64_dpp of a chain (each consumes the accumulator of the one before)
// -- ~100 s_nop per sample on the headline shape.  The hardware needs none: the accumulator is an ordinary
// VALU operand, only `src` is read through DPP, and it is not written inside the chain.  WAIT: two wait
// states first (src may have been produced by the two preceding VALU instructions).
#define ACME_FBS(j) "v_fmac_f64_dpp %[acc], %[src], %[m" #j "] row_newbcast:%[k" #j "] row_mask:0xf bank_mask:0xf\n\t"
#define ACME_FBI(j) [m##j] "v"(mul[OFF + K0 + j]), [k##j] "n"(K0 + j)
#define ACME_FBS_1 ACME_FBS(0)
#define ACME_FBI_1 ACME_FBI(0)
#define ACME_FBS_2 ACME_FBS_1 ACME_FBS(1)
#define ACME_FBI_2 ACME_FBI_1, ACME_FBI(1)
#define ACME_FBS_3 ACME_FBS_2 ACME_FBS(2)
#define ACME_FBI_3 ACME_FBI_2, ACME_FBI(2)
#define ACME_FBS_4 ACME_FBS_3 ACME_FBS(3)
#define ACME_FBI_4 ACME_FBI_3, ACME_FBI(3)
#define ACME_FBS_5 ACME_FBS_4 ACME_FBS(4)
#define ACME_FBI_5 ACME_FBI_4, ACME_FBI(4)
#define ACME_FBS_6 ACME_FBS_5 ACME_FBS(5)
#define ACME_FBI_6 ACME_FBI_5, ACME_FBI(5)
#define ACME_FBS_7 ACME_FBS_6 ACME_FBS(6)
#define ACME_FBI_7 ACME_FBI_6, ACME_FBI(6)
#define ACME_FBS_8 ACME_FBS_7 ACME_FBS(7)
#define ACME_FBI_8 ACME_FBI_7, ACME_FBI(7)
#define ACME_FBS_9 ACME_FBS_8 ACME_FBS(8)
#define ACME_FBI_9 ACME_FBI_8, ACME_FBI(8)
#define ACME_FBS_10 ACME_FBS_9 ACME_FBS(9)
#define ACME_FBI_10 ACME_FBI_9, ACME_FBI(9)
#define ACME_FBS_11 ACME_FBS_10 ACME_FBS(10)
#define ACME_FBI_11 ACME_FBI_10, ACME_FBI(10)
#define ACME_FBS_12 ACME_FBS_11 ACME_FBS(11)
#define ACME_FBI_12 ACME_FBI_11, ACME_FBI(11)
#define ACME_FBS_13 ACME_FBS_12 ACME_FBS(12)
#define ACME_FBI_13 ACME_FBI_12, ACME_FBI(12)
#define ACME_FB_CASE(n)                                                                                   \
    if constexpr (CNT == n) {                                                                             \
        if (WAIT) asm("s_nop 1\n\t" ACME_FBS_##n : [acc] "+v"(acc) : [src] "v"(src), ACME_FBI_##n);       \
        else asm(ACME_FBS_##n : [acc] "+v"(acc) : [src] "v"(src), ACME_FBI_##n);                          \
    }
// CNT (<= 13: the operand limit of an asm statement) multiply-adds for lanes K0 .. K0 + CNT - 1
template <int K0, int CNT, bool WAIT, int OFF, int M> ACME_DEV void fmac_bcast_seg(double &acc, double src, const double (&mul)[M]) {
    static_assert(CNT >= 1 && CNT <= 13 && M >= OFF + K0 + CNT, "");
    ACME_FB_CASE(1) ACME_FB_CASE(2) ACME_FB_CASE(3) ACME_FB_CASE(4) ACME_FB_CASE(5) ACME_FB_CASE(6) ACME_FB_CASE(7)
    ACME_FB_CASE(8) ACME_FB_CASE(9) ACME_FB_CASE(10) ACME_FB_CASE(11) ACME_FB_CASE(12) ACME_FB_CASE(13)
}
template <int N, bool WAIT, int OFF = 0, int M> ACME_DEV void fmac_bcast_chain(double &acc, double src, const double (&mul)[M]) {
    static_assert(N >= 0 && N <= 16, "one DPP row");
    if constexpr (N == 0) {
    } else if constexpr (N <= 13) {
        fmac_bcast_seg<0, N, WAIT, OFF>(acc, src, mul);
    } else {
        fmac_bcast_seg<0, 13, WAIT, OFF>(acc, src, mul);
        fmac_bcast_seg<13, N - 13, false, OFF>(acc, src, mul);
    }
}
// ... with the broadcast lanes L0 .. L0 + N - 1 and the multipliers mul[MB .. MB + N - 1] chosen independently (the
// condensed shapes multiply the reduced unknowns, lanes NL .. NN-1, with a register block indexed from 0)
template <int L0, int N, bool WAIT, int MB = 0, int M> ACME_DEV void fmac_bcast_chain_from(double &acc, double src, const double (&mul)[M]) {
    static_assert(N >= 0 && N <= 13 && L0 >= 0 && L0 + N <= 16 && MB >= 0 && MB + N <= M, "one DPP row, one statement");
    if constexpr (N > 0) fmac_bcast_seg<L0, N, WAIT, MB - L0>(acc, src, mul);
}
#undef ACME_FB_CASE
// The replay of a recorded elimination,  acc += (lane k of acc's row) * mul[k]  for k = 0 .. N-1,  as ONE
// statement: every step reads through DPP what the step before wrote, so each carries its two wait states
// (the hardware's, no interlock); as N statements the compiler added one more after each of them.
#define ACME_FSS(j) "s_nop 1\n\tv_fmac_f64_dpp %[acc], %[acc], %[m" #j "] row_newbcast:%[k" #j "] row_mask:0xf bank_mask:0xf\n\t"
#define ACME_FSS_1 ACME_FSS(0)
#define ACME_FSS_2 ACME_FSS_1 ACME_FSS(1)
#define ACME_FSS_3 ACME_FSS_2 ACME_FSS(2)
#define ACME_FSS_4 ACME_FSS_3 ACME_FSS(3)
#define ACME_FSS_5 ACME_FSS_4 ACME_FSS(4)
#define ACME_FSS_6 ACME_FSS_5 ACME_FSS(5)
#define ACME_FSS_7 ACME_FSS_6 ACME_FSS(6)
#define ACME_FSS_8 ACME_FSS_7 ACME_FSS(7)
#define ACME_FSS_9 ACME_FSS_8 ACME_FSS(8)
#define ACME_FSS_10 ACME_FSS_9 ACME_FSS(9)
#define ACME_FSS_11 ACME_FSS_10 ACME_FSS(10)
#define ACME_FSS_12 ACME_FSS_11 ACME_FSS(11)
#define ACME_FSS_13 ACME_FSS_12 ACME_FSS(12)
#define ACME_FS_CASE(n) \
    if constexpr (CNT == n) asm volatile(ACME_FSS_##n : [acc] "+v"(acc) : ACME_FBI_##n);
template <int K0, int CNT, int M> ACME_DEV void fmac_self_seg(double &acc, const double (&mul)[M]) {
    constexpr int OFF = 0;
    static_assert(CNT >= 1 && CNT <= 13 && M >= K0 + CNT, "");
    ACME_FS_CASE(1) ACME_FS_CASE(2) ACME_FS_CASE(3) ACME_FS_CASE(4) ACME_FS_CASE(5) ACME_FS_CASE(6) ACME_FS_CASE(7)
    ACME_FS_CASE(8) ACME_FS_CASE(9) ACME_FS_CASE(10) ACME_FS_CASE(11) ACME_FS_CASE(12) ACME_FS_CASE(13)
}
template <int N, int M> ACME_DEV void fmac_self_chain(double &acc, const double (&mul)[M]) {
    static_assert(N >= 1 && N <= 16, "one DPP row");
    if constexpr (N <= 13) {
        fmac_self_seg<0, N>(acc, mul);
    } else {
        fmac_self_seg<0, 13>(acc, mul);
        fmac_self_seg<13, N - 13>(acc, mul);
    }
}
// ... replaying the steps of lanes L0 .. L0 + N - 1 with the multipliers mul[0 .. N - 1]
#define ACME_FS_CASE_FROM(n) \
    if constexpr (CNT == n) asm volatile(ACME_FSS_##n : [acc] "+v"(acc) : ACME_FBI_##n);
template <int L0, int N, int M> ACME_DEV void fmac_self_chain_from(double &acc, const double (&mul)[M]) {
    static_assert(N >= 1 && N <= 13 && L0 >= 0 && L0 + N <= 16 && M >= N, "one DPP row, one statement");
    constexpr int K0 = L0, CNT = N, OFF = -L0;
    ACME_FS_CASE_FROM(1) ACME_FS_CASE_FROM(2) ACME_FS_CASE_FROM(3) ACME_FS_CASE_FROM(4) ACME_FS_CASE_FROM(5) ACME_FS_CASE_FROM(6)
    ACME_FS_CASE_FROM(7) ACME_FS_CASE_FROM(8) ACME_FS_CASE_FROM(9) ACME_FS_CASE_FROM(10) ACME_FS_CASE_FROM(11)
    ACME_FS_CASE_FROM(12) ACME_FS_CASE_FROM(13)
}
#undef ACME_FS_CASE_FROM
#undef ACME_FS_CASE

// two wait states before a run of fmac_bcast statements whose source may have just been produced
ACME_DEV void dpp_wait() { asm volatile("s_nop 1"); }
// The scalar head of one Gauss-Jordan step as ONE statement (the compiler brackets every inline-asm
// statement with defensive s_nop's; fused, the step costs 14 issue slots instead of 20):
//   piv  = row_newbcast:K of ak                       (pivot, to every lane of the row)
//   inv  = 1/piv                                      (v_rcp_f64 + the cubic refinement of recip())
//   nlm  = -ak * inv   (0 in the lanes of pivlanes)   (minus the multiplier of every other row)
//   dinv = inv         (in the lanes of pivlanes)     (the pivot row remembers 1/pivot)
//   pivlanes <<= 1 for the next step.  (The |nlm| > 4 ballot stays outside: as an output of a
//   statement that also has vector outputs the compiler would treat the mask as divergent.)
// SAFE: ak may have been written within the two preceding wait states.  Inside: the wait state between the
// transcendental v_rcp_f64 and its first use is the step's own mask shift (into a second scalar pair: the
// unshifted mask is still needed for the narrowed EXEC) instead of an s_nop; EXEC is only written by SALU
// instructions, which neither DPP nor VALU instructions have to wait for.
// One WHOLE Gauss-Jordan step as (at most two) asm statements: the step head above followed by the row updates
//   r[j] += (lane K of r[j]'s row) * nlm      for the CNT registers r[] (the columns right of the pivot, b)
// Between separate statements the compiler assumes a dst-forwarding hazard and inserts a wait state: two per
// step with the head and every update a statement of its own.  Operand limit of a statement: 30 -- the head
// takes 15, every read-modify-write register 2 -- so the first statement carries up to 7 updates, a second
// one the rest.  DPP hazards: r[j] was last written one whole step ago; SAFE (step 0, and wherever ak is fresh):
// two wait states at the head of BOTH statements -- the compiler may copy a register just before either.
#define ACME_FSELF(j) "v_fmac_f64_dpp %[r" #j "], %[r" #j "], %[nlm] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define ACME_RSELF(j) [r##j] "+v"(*rp[O + j])
#define ACME_FSELF_1 ACME_FSELF(0)
#define ACME_RSELF_1 ACME_RSELF(0)
#define ACME_FSELF_2 ACME_FSELF_1 ACME_FSELF(1)
#define ACME_RSELF_2 ACME_RSELF_1, ACME_RSELF(1)
#define ACME_FSELF_3 ACME_FSELF_2 ACME_FSELF(2)
#define ACME_RSELF_3 ACME_RSELF_2, ACME_RSELF(2)
#define ACME_FSELF_4 ACME_FSELF_3 ACME_FSELF(3)
#define ACME_RSELF_4 ACME_RSELF_3, ACME_RSELF(3)
#define ACME_FSELF_5 ACME_FSELF_4 ACME_FSELF(4)
#define ACME_RSELF_5 ACME_RSELF_4, ACME_RSELF(4)
#define ACME_FSELF_6 ACME_FSELF_5 ACME_FSELF(5)
#define ACME_RSELF_6 ACME_RSELF_5, ACME_RSELF(5)
#define ACME_FSELF_7 ACME_FSELF_6 ACME_FSELF(6)
#define ACME_RSELF_7 ACME_RSELF_6, ACME_RSELF(6)
#define ACME_FSELF_8 ACME_FSELF_7 ACME_FSELF(7)
#define ACME_RSELF_8 ACME_RSELF_7, ACME_RSELF(7)
#define ACME_FSELF_9 ACME_FSELF_8 ACME_FSELF(8)
#define ACME_RSELF_9 ACME_RSELF_8, ACME_RSELF(8)
#define ACME_GJ_HEAD_TEXT                                                                          \
        "v_mov_b64_dpp %[piv], %[ak] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"          \
        "v_rcp_f64_e32 %[inv], %[piv]\n\t"                                                       \
        "s_lshl_b64 %[mn], %[m], 1\n\t"                                                          \
        "v_fma_f64 %[e], -%[piv], %[inv], 1.0\n\t"                                               \
        "v_fmac_f64_e32 %[e], %[e], %[e]\n\t"                                                    \
        "v_fmac_f64_e32 %[inv], %[inv], %[e]\n\t"                                                \
        "v_mul_f64 %[nlm], %[ak], -%[inv]\n\t"                                                   \
        "s_and_saveexec_b64 %[sv], %[m]\n\t"                                                     \
        "v_mov_b64 %[dinv], %[inv]\n\t"                                                          \
        "v_mov_b64 %[nlm], 0\n\t"                                                                \
        "v_mov_b64 %[frz], %[vmx]\n\t"                                                           \
        "s_mov_b64 exec, %[sv]\n\t"                                                              \
        "v_max_f64 %[vmx], %[vmx], |%[nlm]|\n\t"
#define ACME_GJ_A(n)                                                                                          \
    if constexpr (NA == n) {                                                                                  \
        constexpr int O = 0;                                                                                  \
        if (SAFE) asm volatile("s_nop 1\n\t" ACME_GJ_HEAD_TEXT ACME_FSELF_##n                                  \
                     : [piv] "=&v"(piv), [inv] "=&v"(inv), [e] "=&v"(e), [nlm] "=&v"(nlm), [dinv] "+v"(dinv),  \
                       [sv] "=&s"(sv), [mn] "=&s"(mnext), [vmx] "+v"(vmx), [frz] "+v"(frz), ACME_RSELF_##n   \
                     : [ak] "v"(ak), [m] "s"(pivlanes), [k] "n"(K) : "scc");                                                      \
        else asm volatile(ACME_GJ_HEAD_TEXT ACME_FSELF_##n                                                     \
                     : [piv] "=&v"(piv), [inv] "=&v"(inv), [e] "=&v"(e), [nlm] "=&v"(nlm), [dinv] "+v"(dinv),  \
                       [sv] "=&s"(sv), [mn] "=&s"(mnext), [vmx] "+v"(vmx), [frz] "+v"(frz), ACME_RSELF_##n   \
                     : [ak] "v"(ak), [m] "s"(pivlanes), [k] "n"(K) : "scc");                                                      \
    }
#define ACME_GJ_B(n)                                                                                          \
    if constexpr (CNT - NA == n) {                                                                            \
        constexpr int O = NA;                                                                                 \
        if (SAFE) asm volatile("s_nop 1\n\t" ACME_FSELF_##n : ACME_RSELF_##n : [nlm] "v"(nlm), [k] "n"(K));     \
        else asm volatile(ACME_FSELF_##n : ACME_RSELF_##n : [nlm] "v"(nlm), [k] "n"(K));                      \
    }
template <int K, int CNT, bool SAFE>
ACME_DEV void gj_step(double ak, double &dinv, unsigned long long &pivlanes, double &nlm, double &vmx, double &frz, double *const (&rp)[CNT]) {
    static_assert(CNT >= 1 && CNT <= 16, "");
    constexpr int NA = CNT < 7 ? CNT : 7;
    double piv, inv, e;
    unsigned long long sv, mnext;
    ACME_GJ_A(1) ACME_GJ_A(2) ACME_GJ_A(3) ACME_GJ_A(4) ACME_GJ_A(5) ACME_GJ_A(6) ACME_GJ_A(7)
    pivlanes = mnext;
    ACME_GJ_B(1) ACME_GJ_B(2) ACME_GJ_B(3) ACME_GJ_B(4) ACME_GJ_B(5) ACME_GJ_B(6) ACME_GJ_B(7) ACME_GJ_B(8) ACME_GJ_B(9)
}
#undef ACME_GJ_A
#undef ACME_GJ_B
#undef ACME_GJ_HEAD_TEXT

// ---- TWO rows per lane (the mid-size kernel, acme_coop.h: 17 ... 32 unknowns, row p in lane p mod 16, slot p / 16) ----
// The head of a Gauss-Jordan step whose pivot row sits in lane K of the slot holding `ak` (its entries of the pivot column:
// ak in the pivot's slot, ao in the other):
//   piv = row_newbcast:K of ak;  inv = 1 / piv (v_rcp_f64 + the cubic refinement of recip());
//   ak <- -ak * inv, ao <- -ao * inv           (minus the multipliers, recorded IN PLACE: column K of the factors)
//   in the pivot's lane: dinv <- inv, ak <- 0, frz <- vmxk      (what gj_step does for one row per lane)
//   vmxk / vmxo: the largest |multiplier| each row has seen (frz: as a row still waiting to be the pivot)
// Always with the two DPP wait states in front: ak was written by the first update of the step before.
#define ACME_GJ2_HEAD_TEXT                                                                          \
        "s_nop 1\n\t"                                                                               \
        "v_mov_b64_dpp %[piv], %[ak] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"              \
        "v_rcp_f64_e32 %[inv], %[piv]\n\t"                                                           \
        "s_nop 0\n\t"                                                                               \
        "v_fma_f64 %[e], -%[piv], %[inv], 1.0\n\t"                                                   \
        "v_fmac_f64_e32 %[e], %[e], %[e]\n\t"                                                        \
        "v_fmac_f64_e32 %[inv], %[inv], %[e]\n\t"                                                    \
        "v_mul_f64 %[ak], %[ak], -%[inv]\n\t"                                                        \
        "v_mul_f64 %[ao], %[ao], -%[inv]\n\t"                                                        \
        "s_and_saveexec_b64 %[sv], %[m]\n\t"                                                         \
        "v_mov_b64 %[dinv], %[inv]\n\t"                                                              \
        "v_mov_b64 %[ak], 0\n\t"                                                                     \
        "v_mov_b64 %[frz], %[vmxk]\n\t"                                                              \
        "s_mov_b64 exec, %[sv]\n\t"                                                                  \
        "v_max_f64 %[vmxk], %[vmxk], |%[ak]|\n\t"                                                    \
        "v_max_f64 %[vmxo], %[vmxo], |%[ao]|\n\t"
template <int K> ACME_DEV void gj2_head(double &ak, double &ao, double &dinv, double &vmxk, double &vmxo, double &frz) {
    double piv, inv, e;
    unsigned long long sv;
    // (the lane mask materialised HERE, as lanes_here() does: hoisted, the 2 x 32 constants of an elimination are spilled)
    unsigned half;
    asm volatile("s_mov_b32 %0, %1" : "=s"(half) : "n"((1u << K) * 0x10001u));
    const unsigned long long m = ((unsigned long long)half << 32) | half;
    asm volatile(ACME_GJ2_HEAD_TEXT
                 : [piv] "=&v"(piv), [inv] "=&v"(inv), [e] "=&v"(e), [ak] "+v"(ak), [ao] "+v"(ao), [dinv] "+v"(dinv),
                   [sv] "=&s"(sv), [vmxk] "+v"(vmxk), [vmxo] "+v"(vmxo), [frz] "+v"(frz)
                 : [m] "s"(m), [k] "n"(K) : "scc");
}
#undef ACME_GJ2_HEAD_TEXT
// ... and its row updates, column by column: first the other slot's row  ro += (lane K of rk's row) * nlo,  then the pivot
// slot's own  rk += (lane K of rk's row) * nlk  (nlk is 0 in the pivot's lane: the pivot row stays).  rk is read through
// DPP before the step writes it, and was last written a whole step ago -- by the program; the register allocator may still
// reload it (v_accvgpr_read: the instantiations live beyond 256 registers) right in front of a statement, so each begins
// with the two wait states (tools/dpp_hazard_check.py found exactly that).  Six columns to a statement (operand limit).
#define ACME_G2S(j) "v_fmac_f64_dpp %[o" #j "], %[p" #j "], %[nlo] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t" \
                    "v_fmac_f64_dpp %[p" #j "], %[p" #j "], %[nlk] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define ACME_G2R(j) [o##j] "+v"(*op[O + j]), [p##j] "+v"(*kp[O + j])
#define ACME_G2S_1 ACME_G2S(0)
#define ACME_G2R_1 ACME_G2R(0)
#define ACME_G2S_2 ACME_G2S_1 ACME_G2S(1)
#define ACME_G2R_2 ACME_G2R_1, ACME_G2R(1)
#define ACME_G2S_3 ACME_G2S_2 ACME_G2S(2)
#define ACME_G2R_3 ACME_G2R_2, ACME_G2R(2)
#define ACME_G2S_4 ACME_G2S_3 ACME_G2S(3)
#define ACME_G2R_4 ACME_G2R_3, ACME_G2R(3)
#define ACME_G2S_5 ACME_G2S_4 ACME_G2S(4)
#define ACME_G2R_5 ACME_G2R_4, ACME_G2R(4)
#define ACME_G2S_6 ACME_G2S_5 ACME_G2S(5)
#define ACME_G2R_6 ACME_G2R_5, ACME_G2R(5)
#define ACME_G2_CASE(n) \
    if constexpr (CNT == n) asm volatile("s_nop 1\n\t" ACME_G2S_##n : ACME_G2R_##n : [nlk] "v"(nlk), [nlo] "v"(nlo), [k] "n"(K));
template <int K, int O, int CNT, int M> ACME_DEV void gj2_update_seg(double nlk, double nlo, double *const (&kp)[M], double *const (&op)[M]) {
    static_assert(CNT >= 1 && CNT <= 6 && O + CNT <= M, "");
    ACME_G2_CASE(1) ACME_G2_CASE(2) ACME_G2_CASE(3) ACME_G2_CASE(4) ACME_G2_CASE(5) ACME_G2_CASE(6)
}
template <int K, int O, int M> ACME_DEV void gj2_update_from(double nlk, double nlo, double *const (&kp)[M], double *const (&op)[M]) {
    if constexpr (O < M) {
        constexpr int CNT = M - O < 6 ? M - O : 6;
        gj2_update_seg<K, O, CNT>(nlk, nlo, kp, op);
        gj2_update_from<K, O + CNT>(nlk, nlo, kp, op);
    }
}
// M columns: kp[j] the pivot slot's register of column j, op[j] the other slot's
template <int K, int M> ACME_DEV void gj2_update(double nlk, double nlo, double *const (&kp)[M], double *const (&op)[M]) {
    gj2_update_from<K, 0>(nlk, nlo, kp, op);
}
#undef ACME_G2_CASE
// The REPLAY of such an elimination on another right-hand side (xk: the entries of the rows in the pivot's slot, xo: the
// other slot's; mk[j] / mo[j]: the recorded multipliers of step K0 + j): every step reads through DPP what the step before
// wrote -- two wait states each.  Eight steps to a statement.
#define ACME_R2S(j) "s_nop 1\n\t" \
                    "v_fmac_f64_dpp %[xo], %[xk], %[mo" #j "] row_newbcast:%[k" #j "] row_mask:0xf bank_mask:0xf\n\t" \
                    "v_fmac_f64_dpp %[xk], %[xk], %[mk" #j "] row_newbcast:%[k" #j "] row_mask:0xf bank_mask:0xf\n\t"
#define ACME_R2I(j) [mo##j] "v"(mo[C0 + j]), [mk##j] "v"(mk[C0 + j]), [k##j] "n"((K0 + j) % 16)
#define ACME_R2S_1 ACME_R2S(0)
#define ACME_R2I_1 ACME_R2I(0)
#define ACME_R2S_2 ACME_R2S_1 ACME_R2S(1)
#define ACME_R2I_2 ACME_R2I_1, ACME_R2I(1)
#define ACME_R2S_3 ACME_R2S_2 ACME_R2S(2)
#define ACME_R2I_3 ACME_R2I_2, ACME_R2I(2)
#define ACME_R2S_4 ACME_R2S_3 ACME_R2S(3)
#define ACME_R2I_4 ACME_R2I_3, ACME_R2I(3)
#define ACME_R2S_5 ACME_R2S_4 ACME_R2S(4)
#define ACME_R2I_5 ACME_R2I_4, ACME_R2I(4)
#define ACME_R2S_6 ACME_R2S_5 ACME_R2S(5)
#define ACME_R2I_6 ACME_R2I_5, ACME_R2I(5)
#define ACME_R2S_7 ACME_R2S_6 ACME_R2S(6)
#define ACME_R2I_7 ACME_R2I_6, ACME_R2I(6)
#define ACME_R2S_8 ACME_R2S_7 ACME_R2S(7)
#define ACME_R2I_8 ACME_R2I_7, ACME_R2I(7)
#define ACME_R2_CASE(n) \
    if constexpr (CNT == n) asm volatile(ACME_R2S_##n : [xk] "+v"(xk), [xo] "+v"(xo) : ACME_R2I_##n);
// steps K0 .. K0 + CNT - 1 (all with their pivot in the SAME slot: K0 / 16 == (K0 + CNT - 1) / 16), multipliers at
// mk / mo [C0 ...]
template <int K0, int CNT, int C0, int M> ACME_DEV void replay2_seg(double &xk, double &xo, const double (&mk)[M], const double (&mo)[M]) {
    static_assert(CNT >= 1 && CNT <= 8 && C0 + CNT <= M && K0 / 16 == (K0 + CNT - 1) / 16, "");
    ACME_R2_CASE(1) ACME_R2_CASE(2) ACME_R2_CASE(3) ACME_R2_CASE(4) ACME_R2_CASE(5) ACME_R2_CASE(6) ACME_R2_CASE(7) ACME_R2_CASE(8)
}
#undef ACME_R2_CASE

// bcast16<K> as a volatile statement (ordered with the fused operations); SAFE: with the two wait
// states built in
template <int K, bool SAFE> ACME_DEV double bcast16_ordered(double v) {
    double r;
    if (SAFE)
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                     : "=v"(r) : "v"(v), "n"(K));
    else
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                     : "=v"(r) : "v"(v), "n"(K));
    return r;
}
// rotate right by R within each 16-lane row (row_ror:R; 32-bit halves, not legal on b64)
template <int R> ACME_DEV double ror16(double v) {
    int lo = dpp_i<0x120 + R>(__double2loint(v));
    int hi = dpp_i<0x120 + R>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// max over the 16 lanes of each row, result in every lane (4 rotate+max steps)
ACME_DEV double allmax16(double v) {
    v = fmax(v, ror16<8>(v));
    v = fmax(v, ror16<4>(v));
    v = fmax(v, ror16<2>(v));
    v = fmax(v, ror16<1>(v));
    return v;
}

// ... of values that are not NaN: v_max_f64 as it is (fmax() quiets a possible signalling NaN first, one more
// v_max_f64 per operand: two of three instructions of a reduction step)
ACME_DEV double max_nn(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
ACME_DEV double allmax16_nn(double v) {
    v = max_nn(v, ror16<8>(v));
    v = max_nn(v, ror16<4>(v));
    v = max_nn(v, ror16<2>(v));
    v = max_nn(v, ror16<1>(v));
    return v;
}

// min / sum over the 16 lanes of each row, result in every lane
// the value lane K of the WAVE holds, in every lane (v_readlane_b32: through the scalar registers -- the mid-size kernel's
// one-instance-per-wave instantiation broadcasts x_k of its triangular sweeps this way)
template <int K> ACME_DEV double lane64(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), K);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), K);
    return __hiloint2double(hi, lo);
}
// the 64-bit value lane `lane` (wave-uniform, a run-time number) of the wave holds, as a scalar
ACME_DEV unsigned long long lanev64(unsigned long long v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}
// a 64-bit value every lane holds alike, as a wave-uniform (scalar) value: the first lane's
ACME_DEV unsigned long long first64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
ACME_DEV double allmin16(double v) {
    v = fmin(v, ror16<8>(v));
    v = fmin(v, ror16<4>(v));
    v = fmin(v, ror16<2>(v));
    v = fmin(v, ror16<1>(v));
    return v;
}
// maximum / minimum over the whole wave, in every lane
ACME_DEV double allmax64(double v) {
    v = allmax16(v);
    return fmax(fmax(lane64<0>(v), lane64<16>(v)), fmax(lane64<32>(v), lane64<48>(v)));
}
ACME_DEV double allmin64(double v) {
    v = allmin16(v);
    return fmin(fmin(lane64<0>(v), lane64<16>(v)), fmin(lane64<32>(v), lane64<48>(v)));
}
ACME_DEV double allsum16(double v) {
    v += ror16<8>(v);
    v += ror16<4>(v);
    v += ror16<2>(v);
    v += ror16<1>(v);
    return v;
}

// arbitrary gather inside a row: value of lane (row base + src) -- ds_bpermute_b32
ACME_DEV int shfl16(int v, int src) {
    int lane = (int)(threadIdx.x & 63);
    return __builtin_amdgcn_ds_bpermute(((lane & ~15) + src) << 2, v);
}
ACME_DEV double shfl16(double v, int src) {
    int lane = (int)(threadIdx.x & 63);
    int addr = ((lane & ~15) + src) << 2;
    int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

ACME_DEV unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// per-lane predicate from a wave-uniform 64-bit lane mask (compile-time constants become two
// s_mov_b32 feeding v_cndmask directly: no v_cmp, no long-lived SGPR pair)
ACME_DEV bool lanes(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
// ... of a constant pattern (the same 16 bits in every row), materialised HERE: for cold code.  Left to itself the
// compiler hoists such constants -- 39 of them for the 13 steps of pivot_order -- to the top of the kernel, cannot
// keep them in scalar registers across the hot loops and spills every one (2 v_writelane at launch, 2 v_readlane per
// use): most of the kernel's "spilled SGPRs".  A volatile statement is neither hoisted nor merged.
template <unsigned ROW16> ACME_DEV bool lanes_here() {
    unsigned half;
    asm volatile("s_mov_b32 %0, %1" : "=s"(half) : "n"((ROW16 & 0xFFFFu) * 0x10001u));
    return __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)half << 32) | half);
}

// 1/x: v_rcp_f64 seed (relative error 2^-24.4, measured: tools/ubench/rcpacc.hip) refined by one
// cubically convergent step  x (1 + e + e^2),  e = 1 - d x  (truncation e^3 = 2^-73): 3 fused
// operations, 1.00 ulp worst case against the correctly rounded reciprocal the reference's inv()
// returns (tools/ubench/rcp3.hip; two quadratic Newton steps need 4 for the same result); no
// div_scale/div_fmas/div_fixup chain on the elimination's critical path (pivots are never
// denormal in practice)
ACME_DEV double recip(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    e = fma(e, e, e);
    return fma(x, e, x);
}
ACME_DEV int ffs32(int v) { return __ffs(v); }
// LDS read-modify-write without a returned value (ds_add_u64 / ds_max_i64): nothing to wait for
ACME_DEV void lds_add(long long *p, long long v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
ACME_DEV void lds_max(long long *p, long long v) {
    (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// A pointer to memory that nothing writes during the launch, about to be read with wave-uniform
// addresses: as a constant-address-space pointer (same addresses as global memory on gfx9) its loads
// are invariant and go through the scalar unit (s_load_*, results in SGPRs) instead of 64 identical
// per-lane vector loads.
template <class T> ACME_DEV const __attribute__((address_space(4))) T *uniform_ro(const T *p) {
    return (const __attribute__((address_space(4))) T *)p;
}

// two neighbouring doubles of LDS (16-byte aligned) with one ds_read_b128
struct pair_t { double lo, hi; };
ACME_DEV pair_t ld2(const double *p) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t v = *reinterpret_cast<const d2_t *>(__builtin_assume_aligned(p, 16));
    return pair_t{v.x, v.y};
}

// ... and the matching 16-byte store (ds_write_b128)
ACME_DEV void st2(double *p, double lo, double hi) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    *reinterpret_cast<d2_t *>(__builtin_assume_aligned(p, 16)) = d2_t{lo, hi};
}

// scheduling fence: nothing is moved across (used to keep a batch of DPP broadcasts ahead of
// the FMAs that consume them: a dependent dpp->fma pair costs ~17 cycles, batched ~9)
ACME_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// streamed host runs (KArgs::u_ready): a counter the HOST advances while the kernel runs, and data a copy engine has
// written since the launch: the counter is read past the caches (system scope)
ACME_DEV long long load_system(const long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
ACME_DEV void acquire_system() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }
// (~25 us: 2 048 waves polling host memory over the bus must not get in the copy engine's way)
ACME_DEV void nap() { for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_sleep(127); }
// optimisation barrier: the value must be materialised here (keeps a speculative computation
// on the near side of a branch instead of being sunk below it)
ACME_DEV double keep(double v) { asm volatile("" : "+v"(v)); return v; }
ACME_DEV int keepi(int v) { asm volatile("" : "+v"(v)); return v; }
// v, two wait states old: a value that DPP instructions may read at once (for sources the compiler's own code has just
// produced -- a select, a copy -- in front of fused chains whose order among themselves is not fixed)
ACME_DEV double settle(double v) { asm volatile("s_nop 1" : "+v"(v)); return v; }
// a use of v here, nothing else (keeps its register allocated up to this point)
ACME_DEV void touch(double v) { asm volatile("" : : "v"(v)); }
// wave-uniform integer the optimiser must not reason about (stops it from cloning a big loop
// body per value of a small state variable)
ACME_DEV int opaque(int v) {
    v = __builtin_amdgcn_readfirstlane(v);   // uniform by construction; tells the compiler so
    asm volatile("" : "+s"(v));
    return v;
}
// wave mask pinned in a scalar register pair here and now: keeps an OR-chain of ballots
// sequential (left to itself the optimiser gathers all terms first and spills them)
ACME_DEV unsigned long long pin(unsigned long long m) { asm volatile("" : "+s"(m)); return m; }
// floating-point literal as a scalar-register operand instead of living in -- and being copied
// between -- vector registers.  Deliberately NOT volatile: volatile asm statements keep their
// program order, which serialised the two interleaved exp() polynomial chains (-5 %).
ACME_DEV double sconst(double v) { asm("" : "+s"(v)); return v; }

// The 16 constants of the junction exponential as ONE table in constant memory, brought into scalar
// registers by two s_load_dwordx16 per call instead of two s_mov_b32 per constant.
//   0 log2(e)  1 ln2_hi  2 ln2_lo  3..13 1/13! .. 1/3!  14 -2100  15 2100
__constant__ double acme_exp_tab[16] = {
    1.4426950408889634, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07,
    2.7557319223985893e-06, 2.48015873015873e-05, 1.984126984126984e-04, 1.388888888888889e-03,
    8.333333333333333e-03, 4.1666666666666664e-02, 1.6666666666666666e-01, -2100.0, 2100.0};
typedef double d8_t __attribute__((ext_vector_type(8)));
struct ExpTab {
    d8_t lo8, hi8;
    ACME_DEV double operator[](int i) const { return i < 8 ? lo8[i] : hi8[i - 8]; }
};
// max(min(k, hi), lo) with scalar-register bounds; as instructions, so that the compiler does not
// add a canonicalising v_max for bounds whose origin (a scalar load) it cannot see
ACME_DEV double clamp_s(double k, double lo, double hi) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(k), "s"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(r), "s"(hi));
    return r;
}
ACME_DEV ExpTab load_exp_tab() {
    ExpTab t;
    const double *p = acme_exp_tab;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t.lo8), "=&s"(t.hi8)    // early-clobber: never overlapping the pointer pair
                 : "s"(p));
    return t;
}

}  // namespace wv
