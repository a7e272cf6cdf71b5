// acme_kernel.h -- batched run!(::DiscreteModel, u) for gfx950: device code.
//
// Mapping (MI355X-first):
//   * one circuit instance = one DPP row = 16 lanes; lane r owns ROW r of every small
//     matrix (Jacobian / LU factors / Jp), element i of every short vector (x, p, z, res).
//     A wavefront carries 4 instances, a 256-thread block 16.
//   * row-local broadcasts (pivot row, x_j, z_j, p_j) are `row_newbcast` DPP moves, the
//     pivot search is a 4-step `row_ror` max-reduction + one wave ballot; only the row
//     interchange of the partially pivoted LU uses ds_bpermute.
//   * the shared model matrices sit in LDS (column-major, so lane r reads row r of a
//     column at consecutive addresses); per-instance state stays in registers for the
//     whole launch and touches HBM once at either end; u / y stream through LDS in
//     coalesced CHUNK-sample tiles.
//   * Newton iteration counts are data dependent: every loop is driven by a wave ballot
//     over per-instance "still active" masks; finished instances idle as masked lanes.
//
// Reference semantics restated here (file:line in the ACME.jl tree):
//   step!            src/ACME.jl:666-715      evaluate!/closures  src/ACME.jl:176-194,236-252
//   SimpleSolver     src/solvers.jl:151-236   HomotopySolver      src/solvers.jl:247-302
//   LinearSolver     src/solvers.jl:38-132    element functions   src/elements.jl (per kind)
//
// The including translation unit must provide namespace wv (cross-lane primitives) and
// ACME_DEV before including this header (acme_wave_hip.h on the GPU).
#pragma once
#include <math.h>

#include <type_traits>

#include "acme_common.h"
#include "acme_slab_layout.h"

#ifndef ACME_LAMBDA
#define ACME_LAMBDA __attribute__((always_inline))
#endif
// branch-layout hints: the rare paths (pivot re-learning, cache hits, failures, the solver-plugin mode of the
// run kernel) out of the fall-through path -- a taken branch costs a wave an instruction-buffer refill
#define ACME_RARE(x) __builtin_expect(!!(x), 0)
#define ACME_USUAL(x) __builtin_expect(!!(x), 1)
#ifndef ACME_DBG  // debug trace hook, only ever defined by the CPU wave emulator
#define ACME_DBG(...)
#endif

namespace acme {

template <int NN_, int NQ_, int NP_, int NX_, int NU_, int NY_, int RARE_ = 0, int NSUB_ = 1, int NL_ = 0> struct Shape {
    static constexpr int NN = NN_, NQ = NQ_, NP = NP_, NX = NX_, NU = NU_, NY = NY_;
    // NL: residual rows that are linear in z for a given p (potentiometer halves; acme_pack.h CondPlan), held by
    // lanes 0 .. NL-1 and CONDENSED out of the Newton system, which then has NR = NN - NL unknowns (z_NL .. z_NN-1;
    // the host permutes the z basis so that the linear rows' pivot columns come first): wave_main "condensed solve"
    static constexpr int NL = NL_, NR = NN_ - NL_;
    static constexpr bool COND = NL_ > 0;
    static constexpr int NE = COND ? NR : NN;      // steps of the elimination the Newton loop runs (and records)
    // NSUB: capacity for nonlinear sub-problems solved one after another each sample
    // (src/ACME.jl:675-697); every sub-problem is padded to (NN, NQ, NP)
    static constexpr int NSUB = NN_ > 0 ? NSUB_ : 0;
    static constexpr int NSUBr = NSUB > 0 ? NSUB : 1;
    // RARE: the MOSFET / tanh op-amp / Jiles-Atherton element functions are compiled in
    static constexpr bool RARE = RARE_ != 0;
    static constexpr int NT = RARE ? 4 : 3;  // max Jq non-zeros of one residual row
    static constexpr int NQS = (NQ + GROUP - 1) / GROUP;  // q rows per lane
    static constexpr int NXS = (NX + GROUP - 1) / GROUP;  // states per lane
    static constexpr int NUR = NU > 0 ? NU : 1;           // prefetch registers per lane
    static constexpr Layout L = make_layout(NN, NQ, NP, NX, NU, NY, RARE_ != 0 ? 4 : 3, NSUBr);
    // The row of [x0 a b c; y0 dy ey fy] a lane applies every sample (NLC doubles), and its row of
    // [dq eq], are the same for the whole launch: the small shapes keep them in REGISTERS, loaded once
    // from the image in HBM (the big shape has no registers to spare and reads them as LDS pairs).
    // That takes NLC + NX + NU LDS reads out of every sample -- and the linear part out of the image
    // a block stages: with 16 private images per block (Monte-Carlo batches) what is left of a
    // fixed-pot superover image is 2.7 KB instead of 7.6, two blocks fit a CU and the 8 192 instances
    // of BASELINE config 4 run in ONE round at two waves per SIMD.
    static constexpr int NLC = 1 + NX + NU + NSUBr * NN;
    static constexpr bool LINREG = NX > 0 && NX + NY <= GROUP && !L.linp && NSUB == 1 && NLC <= 20;
    static constexpr bool DQREG = LINREG && NP > 0 && NLC + NX + NU <= 24;     // (fixed-pot superover: 20 + 12, spills)
    // what a block stages of a PRIVATE image: [IMG0, IMG0 + IMGN) (a shared image is staged whole)
    static constexpr int IMG0 = LINREG ? L.sub0 + (DQREG ? L.pexpr : 0) : 0;
    static constexpr int IMGN = L.total - IMG0;
    // samples staged per coalesced u / y transfer: 16, or 8 for the shapes with many inputs (LDS
    // per block decides whether two blocks fit a CU; a refill exposes ~1 us of HBM latency)
    static constexpr int CH = NU >= 4 ? CHUNK / 2 : CHUNK;
    // per-instance LDS scratch (doubles): u tile | y tile | report words (int64)
    static constexpr int UBUF = CH * NU, YBUF = CH * NY, RBUF = 6;
    static constexpr int SCRATCH = (UBUF + YBUF + RBUF + 2) & ~1;
    // row constants staged in LDS: the kind-by-kind evaluation (RARE) reads all ROWC of them, the
    // unified rows only UR_SA .. UR_W1
    static constexpr int RC0 = RARE ? 0 : UR_SA;
    // RCPAIR (the non-RARE shapes): the 13 constants of a unified row are staged in PAIRS (sA sB | cA cB |
    // dA dB | h - | g0 g1 | g2 w0 | w1 -) and fetched by every evaluate! as 7 16-byte reads instead of
    // living in 14 registers: those registers were re-assigned inside the Newton loop whenever the lanes
    // adopt a new row order, and the compiler paid for that with 26 register copies per Newton pass
    // (+2.1 % on the headline with 25 registers freed, +2.1 % on config 4 -- whose 6 spills go --, +4.6 % on
    // the birdie, +3.2 % on the 16-lane diode clipper)
    static constexpr bool RCPAIR = !RARE;
    static constexpr int ROWC_L = RARE ? ROWC : 14;
    static constexpr int ROWI_L = (ROWI + 1) / 2;            // doubles holding the ROWI ints of a row
    // persistent state per instance: x | last_p of every sub-problem | last_z of every sub-problem
    static constexpr int STATE = NX + NSUBr * (NP + NN);
    // per-wave store of the linearisation at the extrapolation origin, per lane (= per row in the
    // lanes' order at that moment): the NN elimination multipliers of the row, 1/pivot, the NT Jq
    // non-zeros and the NT pfull entries of the row -- what the first-order extrapolated start
    // z0 = last_z - last_J \ (last_Jp (p - last_p))  (src/solvers.jl:209-215) needs, see base_solve.
    // One slab per slot holding the NN rows of each of the wave's 4 instances back to back (lane r
    // keeps row r: consecutive addresses, conflict-free ds_read/write_b64).  LDS per block decides
    // whether 2 blocks (= 2 waves/SIMD) fit a CU, so the slabs are packed to NN rows, not 16.
    // Models with few parameters keep the origin as  J^-1 Jp  (nn x np, row lig per lane) instead:
    // np augmented columns ride along in the elimination of the accepted iterate and the
    // extrapolation is one small mat-vec -- cheaper than recording and replaying the elimination
    // while np is small (measured on MI355X: replaying gains 9 % at np = 11 / nn = 13 and loses
    // 2 % at 5 / 7, 6 % at 3 / 4, 7 % at 1 / 2, where its nn dependent DPP steps dominate).
    static constexpr bool MULT = NP >= 8;
    // Three more per-shape choices (A/B on MI355X, EXPERIMENTS.md; re-measured in round 3 once the fused
    // operations had become single asm statements without compiler-inserted wait states):
    //   FUSE    broadcasts of x_j / z_j / p_j fused into the consuming multiply-add (chains of v_fmac_f64_dpp,
    //           wv::fmac_bcast_chain) instead of v_mov_b64_dpp + v_fmac_f64: headline +1.7 %, fixed-pot superover
    //           (np = 5) +1.9 %, birdie (np = 3) -2.5 %
    //   GJHEAD  the scalar head of an elimination step as one fused asm statement -- pivot lane handled under a
    //           narrowed EXEC instead of four v_cndmask, threshold test as two vector instructions inside it:
    //           headline +3.6 %, fixed-pot superover +2.5 %, birdie +1.4 %
    //   SAFE0   step 0 of the elimination with the two DPP wait states built into every fused operation (the
    //           compiler may copy a row register just before it; it does on the small shapes, never on the big
    //           one -- tools/dpp_hazard_check.py proves which): big +1 % without
    static constexpr bool FUSE = NP >= 5, GJHEAD = NN > 0, SAFE0 = !MULT;
    // solve(solver, p) (acme_batch_solve) as a kernel of its own (wave_main MODE_SOLVE), which takes its
    // pointers and branches out of the run kernel: small shapes +1 .. +4 %; the big one lost 1.5 % in round 2
    // and gains 1.6 % now that the run kernel's rare paths are laid out of line (round 3)
    static constexpr bool SOLVE_SPLIT = true;
    // per-instance element tables (KArgs::table_stride) are compiled in
    static constexpr bool TABLES = NL_ == 0;
    // The homotopy solver's direct attempt as the FIRST PASS of its bisection loop -- one inlined copy of the whole solver
    // stack in the kernel instead of two (round 3 peeled the direct attempt out of the loop: +1.2 ... +12.6 % on kernels
    // with registers to spare).  The condensed kernel has none: with the second copy its run kernel spilled 414 vector
    // registers around the rare paths (428 B of scratch per lane), with one copy 39 (128 B) -- 292.3 -> 275.5 ms per step
    // on the headline, bit-identical (round 5, EXPERIMENTS.md).
#ifdef ACME_EXP_ONELOOP
    static constexpr bool ONELOOP = true;
#else
    static constexpr bool ONELOOP = NL_ > 0;
#endif
    // a launch's iteration total / maximum accumulated in registers by every lane instead of by LDS atomics
    // under a one-lane EXEC: birdie +3.0 %, fixed-pot superover and headline +-0 (the big shape has no
    // registers to spare and keeps the atomics)
    static constexpr bool ITREG = !MULT;
    // every fused broadcast chain waits its two states, not only the first to read a source (and the cache
    // lookup's single fused operations are not used at all): on the shapes whose registers the compiler spills
    // to AGPRs -- the generic ones, with every element kind's code -- a reload (v_accvgpr_read, a VALU write) can land right in
    // front of any of them (tools/dpp_hazard_check.py found one)
    static constexpr bool CHAINWAIT = RARE || NN > 13;
    static constexpr bool ACTM = NN >= 7;
    static constexpr bool EXP2 = NN >= 7 && !RARE;
    // constant lane predicates as literals of the scalar AND (and_rows): small shapes only
    static constexpr bool LITROWS = !MULT;
    // the exponential's 16 constants in vector registers for the whole kernel (the non-RARE shapes have
    // them since their row constants went to LDS): no scalar loads per evaluate!, and 32 of the 102 scalar
    // registers back -- headline +1.5 %, config 4 +2.0 % (with 8 spilled registers), birdie +3.3 %
    // (condensed shapes: the table in SCALAR registers again, fetched per evaluate! -- the condensation occupies 42
    // vector registers for the whole launch, and with the table in 32 more the allocator spilled on the usual path:
    // 1.01 against 1.11e9 on the headline, round 4)
#ifndef ACME_COND_SCALAR_EXP
#define ACME_COND_SCALAR_EXP 1
#endif
    static constexpr bool EXPV = !RARE && !(NL_ > 0 && ACME_COND_SCALAR_EXP);
    static constexpr int OSTRIDE = GROUPS_PER_WAVE * (NN > 0 ? NN : 1);
    static constexpr int OS_MUL = 0, OS_DINV = NE, OS_TV = NE + 1, OS_PF = NE + 1 + NT;
    static constexpr int OSLOTS = MULT ? (NE + 1 + 2 * NT + 1) & ~1 : NP;
    static_assert(!COND || (MULT && NSUB_ == 1 && RARE_ == 0 && NN_ >= 8 && NN_ < GROUP && NL_ < NN_ && NL_ <= 8 && NR <= 13 && NR % 2 == 1),
                  "condensed shapes: one sub-problem of unified rows in the pair layout, row 15 free for the linear lanes' constants");
    static constexpr int ORIGIN1 = OSLOTS * OSTRIDE;          // one sub-problem
    // offset of slot s from the lane's position in the slab.  MULT: slots in PAIRS (two per
    // ds_read_b128 when the extrapolation reads them back), a lane's position counts double
    static constexpr int OPOS = MULT ? 2 : 1;
    ACME_HD static constexpr int oslot(int s) { return MULT ? (s / 2) * 2 * OSTRIDE + (s & 1) : s * OSTRIDE; }
    static constexpr int ORIGIN = NSUBr * ORIGIN1 + GROUP;
    // solution cache of one sub-problem of one instance.  In LDS: cp[NP][CACHE] | count, head;
    // in HBM the same followed by cz[NN][CACHE] (see acme_common.h)
    // LDS: the stored p's (every lookup reads cp[j][lane & 15]: one ds_read_b64 per parameter).  A wave's
    // 32-lane read groups hold two instances, each reading 128 contiguous bytes -- half a bank row -- so
    // two instances must sit an ODD multiple of 128 bytes apart (CACHEI = 16 mod 32 doubles); with the two
    // counters (count, head) in between, neighbours overlapped in 4 banks (round 2: 2-way conflicts on
    // every lookup read).  Single-sub-problem shapes keep the counters in the spare word of the report
    // scratch instead (META_SCR); the others keep them behind the p's and pad.
    static constexpr bool META_SCR = NSUBr == 1;
    static constexpr int CACHEPM = NP * CACHE + 2;            // HBM: p's and the two counters of a sub-problem
    static constexpr int CACHE1 = NP * CACHE + (META_SCR ? 0 : 2);          // LDS doubles per sub-problem
    static constexpr int CACHEI = ((NSUBr * CACHE1 + 15) / 32) * 32 + 16;   // LDS doubles per instance
    static constexpr int CACHE1H = (NP + NN) * CACHE + 2;     // HBM doubles per sub-problem
    static constexpr int CACHEIH = NSUBr * CACHE1H;           // HBM doubles per instance
    // the solution caches sit at the end of the block's LDS and are only allocated (and touched) when
    // the batch runs the caching solver
    // (per_tables: per-instance element tables, KArgs::table_stride -- 16 of them per block instead of one)
    ACME_HD static constexpr int lds_doubles(bool per_instance, bool per_tables = false) {
        return (per_instance ? INST_PER_BLOCK * IMGN : L.total) + (per_tables ? INST_PER_BLOCK : 1) * NSUBr * (ROWC_L * GROUP + ROWI_L * GROUP) +
               INST_PER_BLOCK * SCRATCH + WAVES_PER_BLOCK * ORIGIN;
    }
    // The LOW-LDS kernel variant (wave_main<S, MODE, true>): for batches whose model images -- 16 private
    // ones per block, or the shared one plus the solution caches -- do not fit the 160 KB of a CU.  The
    // image(s) are then READ FROM HBM / L2 where the other variant reads LDS (the kind-by-kind row
    // constants of the RARE shapes too); scratch, origin slabs and solution caches stay in LDS.  Slower
    // (every evaluate! waits for global loads), but no model the shape can hold is refused: the
    // reference derives and runs any model, one by one (src/ACME.jl:150, :650-664).
    static constexpr bool ROWC_G = RARE;      // LOW: row constants from HBM (same layout there as in LDS)
    ACME_HD static constexpr int lds_doubles_low(bool per_tables = false) {
        return (per_tables ? INST_PER_BLOCK : 1) * NSUBr * ((ROWC_G ? 0 : ROWC_L * GROUP) + ROWI_L * GROUP) + INST_PER_BLOCK * SCRATCH +
               WAVES_PER_BLOCK * ORIGIN;
    }
    // Blocks per CU the register budget is cut for (__launch_bounds__): two -- 256 VGPRs per lane -- unless
    // the shape's LDS footprint lets only ONE block (one wave per SIMD) live on a CU anyway: then the whole
    // 512-register file is the wave's (the generic 16-unknown shape spilled 435 registers to scratch under
    // the 256 limit)
    static constexpr int OCC = sizeof(double) * lds_doubles(false, false) > 80 * 1024 ? 1 : 2;
    // shapes that can need it: anything that does not fit with private images and caches
    static constexpr bool HAS_LOW = sizeof(double) * (lds_doubles(true, false) + INST_PER_BLOCK * CACHEI) > 160 * 1024 ||
                                    sizeof(double) * (lds_doubles(false, false) + INST_PER_BLOCK * CACHEI) > 160 * 1024;
    static_assert(sizeof(double) * (lds_doubles_low(false) + INST_PER_BLOCK * CACHEI) <= 160 * 1024, "the LOW-LDS variant must always fit");
};

// compile-time counted loops (indices are template constants: DPP lane selects and
// register-array subscripts must be immediates)
template <int I, int N, class F> ACME_DEV void sfor(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}
template <int I, class F> ACME_DEV void sfor_down(F &&f) {  // I-1 ... 0
    if constexpr (I > 0) {
        f(std::integral_constant<int, I - 1>{});
        sfor_down<I - 1>(f);
    }
}

// compile-time lane predicates on the lane-in-group index (same pattern in all 4 rows)
constexpr unsigned long long rows4(unsigned long long row16) { return (row16 & 0xFFFFull) * 0x0001000100010001ull; }
template <int K> ACME_DEV bool lig_gt() { return wv::lanes(rows4(0xFFFFull << (K + 1))); }        // lig > K
template <int K> ACME_DEV bool lig_eq() { return wv::lanes(rows4(1ull << K)); }                   // lig == K
template <int K, int N> ACME_DEV bool lig_in() {                                                 // K <= lig < N
    return wv::lanes(rows4(((1ull << N) - 1ull) & ~((1ull << K) - 1ull)));
}

// x && (lane-in-group index in the constant set ROWS16): the constant goes into the scalar AND as a
// literal -- written `x && lig < N` the compiler keeps the lane mask of `lig < N` in a scalar register
// pair across the loops, spills it, and reads it back with two v_readlane per use
// (LIT = false: the plain form.  Per shape, measured: small shapes +0.8 ... +2.9 % with the literal, the
// big one -0.9 %)
template <unsigned long long ROWS16, bool LIT> ACME_DEV bool and_rows(bool x) {
    if constexpr (LIT) return wv::lanes(wv::ballot(x) & rows4(ROWS16));
    else return x && wv::lanes(rows4(ROWS16));
}
ACME_DEV double sel(bool c, double a, double b) { return c ? a : b; }

// exp(x) for the junction laws: k = rint(x*log2(e)), r = x - k*ln2 (two-part Cody-Waite),
// degree-13 Taylor polynomial on |r| <= 0.347 (truncation 4e-18 relative), scale by 2^k with
// ldexp (which also saturates to inf / flushes to 0 for out-of-range arguments).  About 1 ulp,
// 19 instructions with two short dependency chains instead of the ~40 of the general-purpose
// library routine.  exp(+-inf) gives NaN here (the library gives inf / 0): both make the residual
// non-finite only in states the solver has already lost.
template <bool SC = true> ACME_DEV double exp_junction(double x, const wv::ExpTab &t) {
    const double k = rint(x * t[0]);
    double r = fma(-k, t[1], x);
    r = fma(-k, t[2], r);
    double p = fma(r, t[3], t[4]);                                                // 1/13!, 1/12!
    // (sconst: keep each coefficient a scalar operand of v_fma_f64 -- the compiler otherwise copies it
    // to vector registers to use the shorter v_fmac encoding, three instructions per step)
    // (SC = false: a table that stays in scalar registers for the whole kernel, as in the lane kernel --
    // there the pinning costs an s_mov and a wait state per coefficient)
    sfor<5, 14>([&](auto ic) ACME_LAMBDA { p = fma(p, r, SC ? wv::sconst(t[decltype(ic)::value]) : t[decltype(ic)::value]); });   // 1/11! .. 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)wv::clamp_s(k, t[14], t[15]));
}
// two exponentials at once: the same arithmetic as exp_junction on each argument, written out
// in lockstep so that the two dependency chains interleave and the constants are loaded once
ACME_DEV void exp_junction2(double xa, double xb, double &ea, double &eb, const wv::ExpTab &t) {
    const double ka = rint(xa * t[0]), kb = rint(xb * t[0]);
    double ra = fma(-ka, t[1], xa), rb = fma(-kb, t[1], xb);
    ra = fma(-ka, t[2], ra);
    rb = fma(-kb, t[2], rb);
    double pa = fma(ra, t[3], t[4]), pb = fma(rb, t[3], t[4]);
    sfor<5, 14>([&](auto ic) ACME_LAMBDA {
        constexpr int i = decltype(ic)::value;
        pa = fma(pa, ra, t[i]);
        pb = fma(pb, rb, t[i]);
    });
    pa = fma(pa, ra, 0.5);
    pb = fma(pb, rb, 0.5);
    pa = fma(pa, ra, 1.0);
    pb = fma(pb, rb, 1.0);
    pa = fma(pa, ra, 1.0);
    pb = fma(pb, rb, 1.0);
    ea = ldexp(pa, (int)wv::clamp_s(ka, t[14], t[15]));
    eb = ldexp(pb, (int)wv::clamp_s(kb, t[14], t[15]));
}
// ... on a table that lives in VECTOR registers for the whole kernel (the same value in every lane):
// no scalar loads per call and none of the 32 scalar registers the table otherwise occupies
struct ExpTabV { double v[16]; };
ACME_DEV void exp_junction2(double xa, double xb, double &ea, double &eb, const ExpTabV &T) {
    const double *t = T.v;
    const double ka = rint(xa * t[0]), kb = rint(xb * t[0]);
    double ra = fma(-ka, t[1], xa), rb = fma(-kb, t[1], xb);
    ra = fma(-ka, t[2], ra);
    rb = fma(-kb, t[2], rb);
    double pa = fma(ra, t[3], t[4]), pb = fma(rb, t[3], t[4]);
    sfor<5, 14>([&](auto ic) ACME_LAMBDA {
        constexpr int i = decltype(ic)::value;
        pa = fma(pa, ra, t[i]);
        pb = fma(pb, rb, t[i]);
    });
    pa = fma(pa, ra, 0.5);
    pb = fma(pb, rb, 0.5);
    pa = fma(pa, ra, 1.0);
    pb = fma(pb, rb, 1.0);
    pa = fma(pa, ra, 1.0);
    pb = fma(pb, rb, 1.0);
    ea = ldexp(pa, (int)fmin(fmax(ka, t[14]), t[15]));
    eb = ldexp(pb, (int)fmin(fmax(kb, t[14]), t[15]));
}
ACME_DEV double exp_junction(double x, const ExpTabV &T) {
    const double *t = T.v;
    const double k = rint(x * t[0]);
    double r = fma(-k, t[1], x);
    r = fma(-k, t[2], r);
    double p = fma(r, t[3], t[4]);
    sfor<5, 14>([&](auto ic) ACME_LAMBDA { p = fma(p, r, t[decltype(ic)::value]); });
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)fmin(fmax(k, t[14]), t[15]));
}
// ... with the 16 constants fetched on the spot (two scalar loads per call)
ACME_DEV double exp_junction(double x) { return exp_junction(x, wv::load_exp_tab()); }
ACME_DEV int sel(bool c, int a, int b) { return c ? a : b; }

// ---------------------------------------------------------------------------------------
// LinearSolver, row-per-lane (src/solvers.jl:46-132).  a[j] = element (lig, j).
// ---------------------------------------------------------------------------------------
template <int NN> struct RowLU {
    // Threshold partial pivoting: the row already sitting in pivot position is kept as long as no
    // later row would need a multiplier larger than this, i.e. as long as |pivot| >= u * max with
    // u = 1/8 -- the relaxed-pivoting trade sparse direct solvers make with u = 0.1 (and the
    // reference's own front end: gensolve, src/ACME.jl:733); the reference's setlhs! is u = 1.  With the
    // strict rule ~5 % of the solves re-learnt the row order only because two candidates of nearly
    // equal size had swapped ranks: +10 % run time for differences at rounding level; 4 -> 8 buys
    // another 4.5 % (16 nothing more), with outputs equal to 12 digits and identical iteration
    // counts on every parity case (tests/solver_pins.py sweeps matrices built to separate the rules).
    static constexpr double PIVOT_THRESHOLD = 8.0;

    // Gauss-Jordan elimination of [A | b | C] in the CURRENT row order, without looking for
    // pivots -- valid whenever the rows already sit in (threshold-)pivot order, which is the
    // normal case because lanes adopt the pivot order of the last factorisation that had to
    // interchange rows (see wave_main).  Row-per-lane makes eliminating above the pivot free (all lanes
    // execute the same FMA anyway), so there is no back-substitution: on return b = A^-1 b
    // (component k in lane k) and c[] = A^-1 C.  Same pivots and multipliers as the
    // reference's setlhs!/solve! (src/solvers.jl:46-132); the upper triangle is eliminated in
    // a different order, i.e. results agree to rounding.  Branch-free; returns a wave mask of
    // the lanes whose multiplier exceeded PIVOT_THRESHOLD (or that got a non-finite result):
    // if the calling instance's bits are set the result is discarded and the caller redoes the
    // job after a partially pivoted factorisation (factor) has told it the pivot order.
    // STORE: the lanes with `keep` RECORD the elimination in their entry of the origin slab -- slot k = minus
    // this row's multiplier of step k (0 in the pivot's own lane), slot NN = 1/pivot of the row -- so that the
    // same linear map can later be applied to another right-hand side (apply_stored): that is all the
    // solver needs of the factorisation at its extrapolation origin.  Written as 16-byte pairs
    // (ds_write_b128; slot s of the entry sits at slab[SH::oslot(s)]) in ONE predicated region at the end:
    // a store per step costs an exec save / restore each, and 8-byte stores at the entries' 16-byte stride
    // are 2-way bank conflicts.  An odd slot count (NN even) leaves 1/pivot over: returned in dinv_out, the
    // caller stores it with the next slot.
    // GJHEAD / SAFE0: see Shape.
    // SAFEALL: every step with the DPP wait states built in (cold callers whose registers the compiler shuffles between
    // the steps: the condensation)
    template <int NC, bool STORE, class SH, bool GJHEAD, bool SAFE0, bool SAFEALL = false>
    static ACME_DEV unsigned long long solve_inplace(double (&a)[NN > 0 ? NN : 1], double &b,
                                                     double (&c)[NC > 0 ? NC : 1], double *slab, bool keep,
                                                     double &dinv_out) {
        unsigned long long viol = 0;
        double dinv = 1.0;   // reciprocal of this lane's pivot
        double rec[STORE ? NN + 1 : 1];
        unsigned long long pivlanes = rows4(1ull);   // lanes holding the pivot row of step k (lig == k)
        // GJHEAD: the largest |multiplier| this row has seen so far (vmx) and -- frozen when the row itself
        // became the pivot row -- the largest it saw as a row BELOW the pivot (frz): the one the threshold is
        // about.  Two vector instructions per step, inside the fused step head, instead of a compare into a
        // scalar pair, two masking s_and_b32 with literals and an s_or_b64 (28 code bytes -> 12): +2.6 %
        double vmx = 0.0, frz = 0.0;
        sfor<0, NN>([&](auto kc) ACME_LAMBDA {
            constexpr int k = decltype(kc)::value;
            // a[k] was written by the first fused operation of step k-1; NN-k-1 more of them, the
            // one on b and NC on c[] followed: two are enough wait states for this DPP read
            constexpr bool far_enough = k > 0 && (NN - k + NC >= 2);
            // row update  a[j] -= l * (pivot row's a[j]),  b and c[] likewise: fused broadcast-FMAs
            // in program order.  Step 0 reads registers the compiler's own code has just written
            // (SAFE forms); from step 1 on, the previous write of each register is at least two
            // of these statements back (see fmac_bcast_self).
            double nlm;                                            // -multiplier of every other row
            if constexpr (GJHEAD) {
                // pivot broadcast, reciprocal, minus the multiplier of every other row (0 for the pivot
                // row itself, which notes 1/pivot in dinv), threshold bookkeeping AND the row updates
                //   a[j] -= l * (pivot row's a[j])  (j > k),  b and c[] likewise
                // as one fused statement (two when there are more than 7 registers to update)
                constexpr int CNT = NN - 1 - k + 1 + NC;
                double *rp[CNT];
                sfor<k + 1, NN>([&](auto jc) ACME_LAMBDA { rp[decltype(jc)::value - k - 1] = &a[decltype(jc)::value]; });
                rp[NN - 1 - k] = &b;
                sfor<0, NC>([&](auto jc) ACME_LAMBDA { rp[NN - k + decltype(jc)::value] = &c[decltype(jc)::value]; });
                wv::gj_step<k, CNT, SAFEALL || !far_enough>(a[k], dinv, pivlanes, nlm, vmx, frz, rp);
            } else {
                const double piv = wv::bcast16_ordered<k, !far_enough>(a[k]);
                const double inv = wv::recip(piv);
                dinv = lig_eq<k>() ? inv : dinv;
                nlm = lig_eq<k>() ? 0.0 : a[k] * -inv;
            }
            unsigned long long big = 0;
            if constexpr (!GJHEAD) big = wv::ballot(fabs(nlm) > PIVOT_THRESHOLD);
            if constexpr (STORE) {
                rec[k] = nlm;
            }
            // rows k+1..NN-1 whose multiplier exceeds the pivot threshold (scalar mask arithmetic)
            if constexpr (!GJHEAD) viol = wv::pin(viol | (big & rows4(((1ull << NN) - 1ull) & ~((2ull << k) - 1ull))));
            if constexpr (!GJHEAD) {
                constexpr bool safe0 = SAFE0 && k == 0;
                sfor<k + 1, NN>([&](auto jc) ACME_LAMBDA {
                    wv::fmac_bcast_self<k, safe0>(a[decltype(jc)::value], nlm);
                });
                wv::fmac_bcast_self<k, safe0>(b, nlm);
                sfor<0, NC>([&](auto jc) ACME_LAMBDA { wv::fmac_bcast_self<k, safe0>(c[decltype(jc)::value], nlm); });
            }
        });
        b *= dinv;
        sfor<0, NC>([&](auto jc) ACME_LAMBDA { c[decltype(jc)::value] *= dinv; });
        if constexpr (STORE) {
            rec[NN] = dinv;
            dinv_out = dinv;
            if (keep)
                sfor<0, (NN + 1) / 2>([&](auto kc) ACME_LAMBDA {
                    constexpr int k = 2 * decltype(kc)::value;
                    wv::st2(&slab[SH::oslot(k)], rec[k], rec[k + 1]);
                });
        }
        // a zero pivot without a larger candidate (exactly singular A) or a NaN in A or b turns every row into
        // NaN; an infinite pivot leaves 1/pivot = 0 behind
        // (one ballot per compare, OR-ed as scalar masks: the ballot of an OR of three booleans comes out as the
        // three compares, a v_cndmask of the OR-ed mask and a fourth compare on that)
        const unsigned long long bad = wv::ballot(!(b * 0.0 == 0.0)) | wv::ballot(dinv == 0.0);
        if constexpr (GJHEAD) viol = wv::ballot(frz > PIVOT_THRESHOLD) | bad;
        else viol |= bad;
        return viol;
    }

    // b <- A^-1 b with the elimination recorded by solve_inplace<.., STORE = true>: the same
    // operations, in the same order, the right-hand side would have seen riding along as an
    // augmented column.  Every step reads, through DPP, the register the previous step wrote: the
    // SAFE forms supply the two wait states.
    // (in two halves, so that the caller can request the entry long before it has the right-hand side)
    template <class SH> static ACME_DEV void load_stored(double (&mul)[NN + 2], const double *slab) {
        sfor<0, (NN + 2) / 2>([&](auto kc) ACME_LAMBDA {       // the row's multipliers and 1/pivot, two slots per LDS read
            constexpr int k = 2 * decltype(kc)::value;
            const wv::pair_t v = wv::ld2(&slab[SH::oslot(k)]);
            mul[k] = v.lo;
            mul[k + 1] = v.hi;
        });
    }
    static ACME_DEV void apply_loaded(double &b, const double (&mul)[NN + 2]) {
        wv::fmac_self_chain<NN>(b, mul);
        b *= mul[NN];
    }

    // ---- condensed shapes: the same elimination in two RANGES of steps ---------------------------------
    // Steps K0 .. K1-1 of the Gauss-Jordan elimination above (pivot rows in the lanes K0 .. K1-1, pivot columns of
    // the same numbers) on a row of NN entries: every entry right of the pivot column is updated -- the columns
    // K1 .. NN-1 ride along as augmented columns -- and so does every lane, whether its row belongs to the range or
    // not.  The condensed shapes run the elimination of a Jacobian whose rows 0 .. NL-1 do not depend on z in two
    // parts: <0, NL> once per change of the potentiometer positions (wave_main: condense) and <NL, NN> in the
    // Newton loop, with the lanes 0 .. NL-1 riding along (their multipliers are not subject to the pivot
    // threshold: those rows are "above the pivot").  STORE: as in solve_inplace, K1 - K0 multipliers and 1 / pivot.
    template <int K0, int K1, int NC, bool STORE, class SH, bool SAFEALL = false>
    static ACME_DEV unsigned long long solve_range(double (&a)[NN > 0 ? NN : 1], double &b, double (&c)[NC > 0 ? NC : 1],
                                                   double *slab, bool keep, double &dinv_out) {
        constexpr int KN = K1 - K0;
        static_assert(K0 >= 0 && K1 <= NN && KN >= 1 && (!STORE || (KN + 1) % 2 == 0), "");
        double dinv = 1.0;
        double rec[STORE ? KN + 1 : 1];
        unsigned long long pivlanes = rows4(1ull << K0);
        double vmx = 0.0, frz = 0.0;
        sfor<K0, K1>([&](auto kc) ACME_LAMBDA {
            constexpr int k = decltype(kc)::value;
            constexpr bool far_enough = k > K0 && (NN - k + NC >= 2);
            constexpr int CNT = NN - 1 - k + 1 + NC;
            double nlm;
            double *rp[CNT];
            sfor<k + 1, NN>([&](auto jc) ACME_LAMBDA { rp[decltype(jc)::value - k - 1] = &a[decltype(jc)::value]; });
            rp[NN - 1 - k] = &b;
            sfor<0, NC>([&](auto jc) ACME_LAMBDA { rp[NN - k + decltype(jc)::value] = &c[decltype(jc)::value]; });
            wv::gj_step<k, CNT, SAFEALL || !far_enough>(a[k], dinv, pivlanes, nlm, vmx, frz, rp);
            if constexpr (STORE) rec[k - K0] = nlm;
        });
        b *= dinv;
        sfor<0, NC>([&](auto jc) ACME_LAMBDA { c[decltype(jc)::value] *= dinv; });
        dinv_out = dinv;
        if constexpr (STORE) {
            rec[KN] = dinv;
            if (keep)
                sfor<0, (KN + 1) / 2>([&](auto kc) ACME_LAMBDA {
                    constexpr int k = 2 * decltype(kc)::value;
                    wv::st2(&slab[SH::oslot(k)], rec[k], rec[k + 1]);
                });
        }
        const unsigned long long bad = wv::ballot(!(b * 0.0 == 0.0)) | wv::ballot(dinv == 0.0);
        return wv::ballot(frz > PIVOT_THRESHOLD) | bad;
    }
    // the recorded steps of lanes K0 .. K0 + KN - 1 applied to another right-hand side (mul: KN multipliers, 1 / pivot)
    template <int KN, class SH> static ACME_DEV void load_stored_n(double (&mul)[KN + 1], const double *slab) {
        static_assert((KN + 1) % 2 == 0, "");
        sfor<0, (KN + 1) / 2>([&](auto kc) ACME_LAMBDA {
            constexpr int k = 2 * decltype(kc)::value;
            const wv::pair_t v = wv::ld2(&slab[SH::oslot(k)]);
            mul[k] = v.lo;
            mul[k + 1] = v.hi;
        });
    }
    template <int K0, int KN> static ACME_DEV void apply_loaded_from(double &b, const double (&mul)[KN + 1]) {
        wv::fmac_self_chain_from<K0, KN>(b, mul);
        b *= mul[KN];
    }
    // pivot_order (below) restricted to the rows / columns K0 .. K1-1: the other lanes keep their rows
    template <int K0, int K1> static ACME_DEV bool pivot_order_range(double (&a)[NN > 0 ? NN : 1], int &orig, int lig, int grp) {
        bool ok = true;
        orig = lig;
        sfor<K0, K1>([&](auto kc) ACME_LAMBDA {
            constexpr int k = decltype(kc)::value;
            const bool in_k = wv::lanes_here<((1u << K1) - 1u) & ~((1u << k) - 1u)>();
            const bool eq_k = wv::lanes_here<1u << k>();
            const bool gt_k = wv::lanes_here<((1u << K1) - 1u) & ~((2u << k) - 1u)>();
            double v = in_k ? fabs(a[k]) : -1.0;
            double m = wv::allmax16(v);
            unsigned long long bal = wv::ballot(v == m);
            int msk = (int)((bal >> (grp * GROUP)) & (unsigned long long)(((1u << K1) - 1u) & ~((1u << k) - 1u)));
            int kp = msk ? wv::ffs32(msk) - 1 : k;          // first row holding the maximum (all NaN: no interchange)
            int src = eq_k ? kp : ((lig == kp) ? k : lig);   // interchange rows k <-> kp
            sfor<k, K1>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                a[j] = wv::shfl16(a[j], src);
            });
            orig = wv::shfl16(orig, src);
            double piv = wv::bcast16<k>(a[k]);
            ok = ok && (piv != 0.0);
            double lm = gt_k ? a[k] * wv::recip(piv) : 0.0;
            sfor<k + 1, K1>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                a[j] = fma(-lm, wv::bcast16<k>(a[j]), a[j]);
            });
        });
        return ok;
    }

    // setlhs! with partial pivoting (first strict maximum, src/solvers.jl:58-78), run only to
    // LEARN the pivot order when solve_inplace reported a violation (a few % of the solves):
    // `orig` returns the original row now stored in this lane (the composed row interchanges);
    // the factors are not kept -- the trailing sub-matrix is all the next pivot search needs.
    // Straight-line code (always search, always interchange).  Returns false for an exactly
    // singular matrix.
    static ACME_DEV bool pivot_order(double (&a)[NN > 0 ? NN : 1], int &orig, int lig, int grp) {
        bool ok = true;
        orig = lig;
        sfor<0, NN>([&](auto kc) ACME_LAMBDA {
            constexpr int k = decltype(kc)::value;
            // (nn >= 7: the lane predicates are materialised on the spot, wv::lanes_here -- hoisted to the top of the
            // kernel, the 3 x nn constants were 430 of the headline kernel's 483 spilled scalar registers; the
            // birdie's nn = 4 kernel measures 1 % slower with it and keeps the hoisted form)
            constexpr bool HERE = NN >= 7;
            auto in_k = [&]() ACME_LAMBDA { if constexpr (HERE) return wv::lanes_here<((1u << NN) - 1u) & ~((1u << k) - 1u)>(); else return lig_in<k, NN>(); };
            auto eq_k = [&]() ACME_LAMBDA { if constexpr (HERE) return wv::lanes_here<1u << k>(); else return lig_eq<k>(); };
            auto gt_k = [&]() ACME_LAMBDA { if constexpr (HERE) return wv::lanes_here<(0xFFFFu << (k + 1)) & 0xFFFFu>(); else return lig_gt<k>(); };
            double v = in_k() ? fabs(a[k]) : -1.0;
            double m = wv::allmax16(v);
            unsigned long long bal = wv::ballot(v == m);
            int msk = (int)((bal >> (grp * GROUP)) & 0xFFFFull);
            int kp = wv::ffs32(msk) - 1;          // first row holding the maximum
            int src = eq_k() ? kp : ((lig == kp) ? k : lig);   // interchange rows k <-> kp
            sfor<k, NN>([&](auto jc) ACME_LAMBDA {  // columns < k are dead
                constexpr int j = decltype(jc)::value;
                a[j] = wv::shfl16(a[j], src);
            });
            orig = wv::shfl16(orig, src);
            double piv = wv::bcast16<k>(a[k]);
            ok = ok && (piv != 0.0);
            double lm = gt_k() ? a[k] * wv::recip(piv) : 0.0;   // l_ik, 0 on rows <= k
            sfor<k + 1, NN>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                a[j] = fma(-lm, wv::bcast16<k>(a[j]), a[j]);
            });
        });
        return ok;
    }

};

// ---------------------------------------------------------------------------------------
// element nonlinearities, one residual row per lane (src/elements.jl)
// ---------------------------------------------------------------------------------------
struct RowDesc {
    int kind, erow, flags;
    double k[8];       // the row constants the common kinds use, cached in registers
    const double *rc;  // all row constants in LDS: rc[c * GROUP]
};

ACME_DEV double rcv(const RowDesc &rd, int c) { return rd.rc[c * GROUP]; }

// Residual and Jq non-zeros of this lane's row.  e[t] is q[tc[t]] (the q entries the row
// depends on, in the order of its Jq non-zeros), tv[t] the matching derivative; exA/exB are
// the hoisted exponentials exp(e[0]*k[0]), exp(e[1]*k[1]).
//   diode (v, i) | bjt row0 (vE, vC, iE), row1 (vE, vC, iC) | pot row0 (v1, i1, pos),
//   row1 (v2, i2, pos) | mosfet (vgs, vds, id) | tanh op-amp (vi, vo) | JA (q1..q4)
// Branch-free evaluation of the common kinds (diode, Ebers-Moll BJT, potentiometer, padding rows)
// from per-row constants -- see UnifiedRowConst in acme_common.h.  Every lane runs the same 16
// instructions whatever element its row belongs to; the kind-by-kind version below costs ~60
// instructions in nested exec-masked regions.  xA/xB are the exponentials' arguments (0 for
// rows without a junction: exp(0) - 1 = 0).
// ... with the row's 13 constants handed over (c[i] = constant UR_SA + i)
template <int NT>
ACME_DEV void eval_row_unified_c(const double (&c)[14], const double (&e)[NT], double exA, double exB,
                                 double &res, double (&tv)[NT]) {
    const double cA = c[UR_CA - UR_SA], cB = c[UR_CB - UR_SA], dA = c[UR_DA - UR_SA], dB = c[UR_DB - UR_SA],
                 h = c[UR_H - UR_SA];
    const double g0 = c[UR_G0 - UR_SA], g1 = c[UR_G1 - UR_SA], g2 = c[UR_G2 - UR_SA], w0 = c[UR_W0 - UR_SA],
                 w1 = c[UR_W1 - UR_SA];
    const double hw = h * fma(w1, e[2], w0);
    double r = cA * (exA - 1.0);
    r = fma(cB, exB - 1.0, r);
    r = fma(g0, e[0], r);
    r = fma(g1, e[1], r);
    r = fma(g2, e[2], r);
    res = fma(hw, e[1], r);
    tv[0] = fma(dA, exA, g0);
    tv[1] = fma(dB, exB, g1 + hw);
    tv[2] = fma(h, e[1], g2);
    for (int t = 3; t < NT; ++t) tv[t] = 0.0;
}
template <bool RARE, int NT>
ACME_DEV void eval_row(const RowDesc &rd, const double (&e)[NT], double exA, double exB,
                       double &res, double (&tv)[NT]) {
    res = 0.0;
    for (int t = 0; t < NT; ++t) tv[t] = 0.0;
    const int kind = rd.kind;
    if (kind == RK_DIODE) {  // src/elements.jl:238-244
        res = rd.k[1] * (exA - 1.0) - e[1];
        tv[0] = rd.k[2] * exA;
        tv[1] = -1.0;
    } else if (kind == RK_BJT) {  // src/elements.jl:323-401
        double vE = e[0], vC = e[1];
        double expE = exA, expC = exB;
        double i_f = rd.k[2] * (expE - 1.0);
        double i_r = rd.k[3] * (expC - 1.0);
        double di_f1 = rd.k[4] * expE;
        double di_r2 = rd.k[5] * expC;
        double i_cc = i_f - i_r, di_cc1 = di_f1, di_cc2 = -di_r2;
        double iBE = rd.k[6] * i_f, diBE1 = rd.k[6] * di_f1;
        double iBC = rd.k[7] * i_r, diBC2 = rd.k[7] * di_r2;
        if (RARE) {  // Gummel-Poon refinements (Early effect, high-level injection, leakage)
            const int fl = rd.flags;
            if (fl & (RF_EARLY | RF_KNEE)) {
                if (!(fl & RF_KNEE)) {  // Early effect only (:335-343)
                    double q1i = 1.0 - vE * rcv(rd, 8) - vC * rcv(rd, 9);
                    i_cc = q1i * (i_f - i_r);
                    di_cc1 = rcv(rd, 18) * (i_f - i_r) + q1i * di_f1;
                    di_cc2 = rcv(rd, 19) * (i_f - i_r) - q1i * di_r2;
                } else if (!(fl & RF_EARLY)) {  // high-level injection only (:344-356)
                    double q2 = i_f * rcv(rd, 10) + i_r * rcv(rd, 11);
                    double qden = 1.0 + sqrt(1.0 + 4.0 * q2);
                    double qfact = 2.0 / qden;
                    i_cc = qfact * (i_f - i_r);
                    double dq21 = di_f1 * rcv(rd, 10), dq22 = di_r2 * rcv(rd, 11);
                    double dqfact1 = -4.0 * dq21 / (qden - 1.0) / (qden * qden);
                    double dqfact2 = -4.0 * dq22 / (qden - 1.0) / (qden * qden);
                    di_cc1 = dqfact1 * (i_f - i_r) + qfact * di_f1;
                    di_cc2 = dqfact2 * (i_f - i_r) - qfact * di_r2;
                } else {  // both (:357-373)
                    double q1i = 1.0 - vE * rcv(rd, 8) - vC * rcv(rd, 9);
                    double q2 = i_f * rcv(rd, 10) + i_r * rcv(rd, 11);
                    double qden = 1.0 + sqrt(1.0 + 4.0 * q2);
                    double qfact = 2.0 * q1i / qden;
                    i_cc = qfact * (i_f - i_r);
                    double dq21 = di_f1 * rcv(rd, 10), dq22 = di_r2 * rcv(rd, 11);
                    double dqfact1 = (2.0 * rcv(rd, 18) * qden - q1i * 4.0 * dq21 / (qden - 1.0)) / (qden * qden);
                    double dqfact2 = (2.0 * rcv(rd, 19) * qden - q1i * 4.0 * dq22 / (qden - 1.0)) / (qden * qden);
                    di_cc1 = dqfact1 * (i_f - i_r) + qfact * di_f1;
                    di_cc2 = dqfact2 * (i_f - i_r) - qfact * di_r2;
                }
            }
            if (fl & RF_ILE) {  // :377-385
                double expEl = (fl & RF_ETAEL) ? exp(vE * rcv(rd, 14)) : expE;
                iBE += rcv(rd, 12) * (expEl - 1.0);
                diBE1 += rcv(rd, 16) * expEl;
            }
            if (fl & RF_ILC) {  // :388-396
                double expCl = (fl & RF_ETACL) ? exp(vC * rcv(rd, 15)) : expC;
                iBC += rcv(rd, 13) * (expCl - 1.0);
                diBC2 += rcv(rd, 17) * expCl;
            }
        }
        tv[2] = -1.0;
        if (rd.erow == 0) {
            res = i_cc + iBE - e[2];
            tv[0] = di_cc1 + diBE1;
            tv[1] = di_cc2;
        } else {
            res = -i_cc + iBC - e[2];
            tv[0] = -di_cc1;
            tv[1] = -di_cc2 + diBC2;
        }
    } else if (kind == RK_POT) {  // src/elements.jl:25-30: (v, i, pos) of this half
        double r = rd.k[0];
        double w = (rd.erow == 0) ? e[2] : (1.0 - e[2]);
        double rw = r * w;
        res = e[0] - rw * e[1];
        tv[0] = 1.0;
        tv[1] = -rw;
        tv[2] = -r * e[1];
    } else if (kind == RK_PAD) {  // host-side shape padding: res = q, keeps z_pad = 0
        res = e[0];
        tv[0] = 1.0;
    } else if (RARE) {
        if (kind == RK_MOSFET) {  // src/elements.jl:453-479
            double pol = rcv(rd, 0), lam = rcv(rd, 1);
            int nvt = (int)rcv(rd, 2), na = (int)rcv(rd, 7);
            double vgs = e[0], vds = e[1], id = e[2];
            double xg = pol * vgs;
            // Horner (Base.evalpoly); derivative coefficients k*c_k at rc[12..14], rc[15..17]
            double a_ = rcv(rd, 8 + na - 1), da = (na > 1) ? rcv(rd, 15 + na - 2) : 0.0;
            for (int k = na - 2; k >= 0; --k) a_ = a_ * xg + rcv(rd, 8 + k);
            for (int k = na - 3; k >= 0; --k) da = da * xg + rcv(rd, 15 + k);
            double vt_ = rcv(rd, 3 + nvt - 1), dvt_ = (nvt > 1) ? rcv(rd, 12 + nvt - 2) : 0.0;
            for (int k = nvt - 2; k >= 0; --k) vt_ = vt_ * xg + rcv(rd, 3 + k);
            for (int k = nvt - 3; k >= 0; --k) dvt_ = dvt_ * xg + rcv(rd, 12 + k);
            double lam_ = vds >= 0.0 ? lam : 0.0;
            tv[2] = -1.0;
            if (vgs <= vt_) {
                res = -id;
            } else if (vds <= vgs - vt_) {
                res = a_ * (vgs - vt_ - 0.5 * vds) * vds * (1.0 + lam_ * vds) - id;
                tv[0] = a_ * (1.0 - dvt_) * vds * (1.0 + lam_ * vds) +
                        da * (vgs - vt_ - 0.5 * vds) * vds * (1.0 + lam_ * vds);
                tv[1] = a_ * (vgs - vt_ + vds * (2.0 * lam_ * (vgs - vt_ - 0.75 * vds) - 1.0));
            } else {
                double d = vgs - vt_;
                res = (a_ / 2.0) * (d * d) * (1.0 + lam_ * vds) - id;
                tv[0] = a_ * d * (1.0 - dvt_) * (1.0 + lam_ * vds) + da / 2.0 * (d * d) * (1.0 + lam_ * vds);
                tv[1] = lam_ * a_ / 2.0 * (d * d);
            }
        } else if (kind == RK_MACAK) {  // src/elements.jl:540-546
            double gain = rcv(rd, 0), scale = rcv(rd, 1);
            double vs = e[0] * rcv(rd, 2);  // gain / scale
            double ch = cosh(vs);
            res = tanh(vs) * scale - e[1];
            tv[0] = gain / (ch * ch);
            tv[1] = -1.0;
        } else if (kind == RK_JA) {  // src/elements.jl:107-129
            double Ms = rcv(rd, 0), al = rcv(rd, 2), c = rcv(rd, 3), k = rcv(rd, 4);
            double s = rcv(rd, 5);    // 1e-4 / Ms
            double cMa = rcv(rd, 6);  // c * Ms / a
            double q1 = e[0], q2 = e[1], q3 = e[2], q4 = e[NT - 1];
            double coth = 1.0 / tanh(q1);
            double aq1 = fabs(q1);
            double Lq = aq1 < 1e-4 ? q1 / 3.0 : coth - 1.0 / q1;
            double Ld = aq1 < 1e-4 ? 1.0 / 3.0 : 1.0 / (q1 * q1) - coth * coth + 1.0;
            double Ld2 = aq1 < 1e-3 ? -2.0 / 15.0 * q1 : 2.0 * coth * (coth * coth - 1.0) - 2.0 / (q1 * q1 * q1);
            double delta = q3 > 0.0 ? 1.0 : -1.0;
            double Man = Ms * Lq;
            double d = Man - q2;
            int s3 = (q3 > 0.0) - (q3 < 0.0), sd = (d > 0.0) - (d < 0.0);
            double dM = (s3 == sd) ? 1.0 : 0.0;
            double den = delta * (k * (1.0 - c)) - al * d;
            res = s * ((1.0 - c) * dM * d / den * q3 + cMa * (q3 + al * q4) * Ld - q4);
            tv[0] = s * (((1.0 - c) * (1.0 - c) * k * Ms) * dM * Ld * delta / (den * den) * q3 +
                         cMa * (q3 + al * q4) * Ld2);
            tv[1] = s * -((1.0 - c) * (1.0 - c)) * k * dM * delta / (den * den) * q3;
            tv[2] = s * ((1.0 - c) * dM * d / den + cMa * Ld);
            tv[NT - 1] = s * (rcv(rd, 7) * Ld - 1.0);  // c*Ms/a*alpha
        }
    }
}

// ---------------------------------------------------------------------------------------
// the per-wave time loop
// ---------------------------------------------------------------------------------------
// MODE_RUN: run! (the hot kernel).  MODE_SOLVE: ONE solve(solver, p) per instance, the solver-plugin
// contract (acme_batch_solve) -- the same code with the time loop cut down to the solve, as a kernel of
// its own so that its pointers and branches stay out of the hot one.  MODE_JAC: one pass that exports every instance's
// get_extrapolation_jacobian(solver) = -(J \ Jp) at its extrapolation origin (src/solvers.jl:198-201),
// a separate, small kernel so that its extra registers and code stay out of the hot one.
// MODE_RUN_STREAM: run! whose u is still being copied into HBM while the kernel runs (KArgs::u_ready; streamed
// host-buffer runs) -- a variant of its own: the check in the tile fetch cost the lone waves of BASELINE config 5 3.9 %.
enum { MODE_RUN = 0, MODE_JAC = 1, MODE_SOLVE = 2, MODE_RUN_STREAM = 3 };
// the batch instance slot `slot` of a launch works on (KArgs::inst_map), clamped into the batch for the empty slots of an
// incomplete last block: what a block stages for them does not matter, but it must be readable
ACME_DEV long long slot_instance(const KArgs &A, long long slot) {
    long long ii = slot < A.n_inst ? slot : A.n_inst - 1;
    if (A.inst_map) ii = A.inst_map[ii];
    return ii < 0 ? 0 : ii;
}
ACME_DEV long long block_instance(const KArgs &A, int g) { return slot_instance(A, (long long)wv::bid() * INST_PER_BLOCK + g); }
template <class S, int MODE = MODE_RUN, bool LOW = false> ACME_DEV void wave_main(const KArgs &A, double *lds) {
    constexpr int NN = S::NN, NQ = S::NQ, NP = S::NP, NX = S::NX, NU = S::NU, NY = S::NY;
    constexpr int NQS = S::NQS, NXS = S::NXS, NT = S::NT, NSUB = S::NSUBr;
    constexpr int NNr = NN > 0 ? NN : 1, NPr = NP > 0 ? NP : 1, NQSr = NQS > 0 ? NQS : 1,
                  NXSr = NXS > 0 ? NXS : 1;
    constexpr Layout L = S::L;
    using LU = RowLU<NN>;
    // condensed solve (Shape::NL > 0): everything but the Jacobian export, which keeps the full system
    constexpr bool COND = S::COND && MODE != MODE_JAC;
    constexpr int NL = S::NL, NR = S::NR, NRr = COND ? NR : 1, NLr = COND ? NL : 1;
    using LUL = RowLU<COND ? NL : 1>;       // the NL x NL block of the linear rows

    const int tid = wv::tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int lig = lane & (GROUP - 1), grp = lane >> 4;
    const int gib = wave * GROUPS_PER_WAVE + grp;  // group in block
    const long long slot = (long long)wv::bid() * INST_PER_BLOCK + gib;
    const bool per_inst = A.image_stride != 0;
    // (a launch over a subset of the batch, or with the instances placed by the launcher: KArgs::inst_map -- slot ->
    // instance, -1 for a slot left empty: a batch too small to give every SIMD its waves is spread thin)
    const long long mapped = (A.inst_map && slot < A.n_inst) ? (long long)A.inst_map[slot] : slot;
    const bool valid = slot < A.n_inst && mapped >= 0;
    const long long inst = valid ? mapped : slot;

    // ---- LDS carve-up -------------------------------------------------------------------
    lds = static_cast<double *>(__builtin_assume_aligned(lds, 16));   // the block's dynamic LDS starts at 0
    double *lds_img = lds;
    constexpr bool ROWC_G = LOW && S::ROWC_G;        // LOW (Shape::lds_doubles_low): no image in LDS, ...
    // element tables: one for the block, or one per instance of the block (KArgs::table_stride)
    // (not in the condensed shapes: the mere presence of the per-instance path cost the headline kernel 0.8 % -- 279.2
    // against 276.9 ms, ten more spilled registers; acme_batch_set_matrices refuses such batches, Shape::TABLES)
#ifdef ACME_EXP_NO_INSTANCE_TABLES      // (developer A/B: the kernel without per-instance element tables)
    constexpr bool per_tab = false;
#else
    const bool per_tab = S::TABLES && A.table_stride != 0;
#endif
    const int tabs = per_tab ? INST_PER_BLOCK : 1;
    double *lds_rowc = lds_img + (LOW ? 0 : per_inst ? INST_PER_BLOCK * S::IMGN : L.total);   // [tabs][NSUB][ROWC_L*16]
    int *lds_rowi = (int *)(lds_rowc + (ROWC_G ? 0 : tabs * NSUB * S::ROWC_L * GROUP));        // [tabs][NSUB][ROWI*16]
    double *lds_scr = reinterpret_cast<double *>(lds_rowi) + tabs * NSUB * S::ROWI_L * GROUP;
    constexpr int OS = S::OSTRIDE;  // slab stride; only lanes lig < NN may store
    // this lane's entry in the wave's origin slab: MULT shapes read / write it as 16-byte pairs, at positions
    // chosen so that neither the ds_read_b128 nor the ds_write_b128 lane groups collide (acme_slab_layout.h)
    double *const ojp0 = lds_scr + INST_PER_BLOCK * S::SCRATCH + wave * S::ORIGIN +
                         (S::MULT ? 2 * (int)SLAB_POS[NN][lane] : grp * NN + lig);
    double *const cache0 = lds_scr + INST_PER_BLOCK * S::SCRATCH + WAVES_PER_BLOCK * S::ORIGIN + gib * S::CACHEI;
    // context of the sub-problem being solved (switched by enter_sub)
    double *ojp = ojp0;            // origin's J^-1 * Jp, row lig: [j * OS]
    double *cch = cache0;          // solution cache of the current sub-problem: stored p's (LDS)
    double *czg = A.cache + (valid ? inst : 0) * S::CACHEIH + S::CACHEPM;  // ... and stored z's (HBM)
    // this instance's tables (its own, or the block's)
    const double *const rowc_b = ROWC_G ? A.rowc + (per_tab && valid ? inst * A.table_stride : 0)
                                        : lds_rowc + (per_tab ? gib : 0) * (NSUB * S::ROWC_L * GROUP);
    const int *const rowi_b = lds_rowi + (per_tab ? gib : 0) * (NSUB * ROWI * GROUP);
    const double *rowc_s = rowc_b;
    const int *rowi_s = rowi_b;
    {   // cooperative load of the model image(s) and the row tables
        const int nthreads = WAVES_PER_BLOCK * 64;
        if constexpr (LOW) {
            // nothing to stage: the image(s) are read from HBM
        } else if (!per_inst) {
            for (int i = tid; i < L.total; i += nthreads) lds_img[i] = A.image[i];
        } else {
            for (int g = 0; g < INST_PER_BLOCK; ++g) {
                long long ii = (long long)wv::bid() * INST_PER_BLOCK + g;
                if (ii >= A.n_inst) ii = A.n_inst - 1;
                if (A.inst_map) ii = A.inst_map[ii];
                if (ii < 0) ii = 0;              // (an empty slot: any image will do)
                const double *src = A.image + ii * A.image_stride;
                for (int i = tid; i < S::IMGN; i += nthreads) lds_img[g * S::IMGN + i] = src[S::IMG0 + i];
            }
        }
        for (int g = 0; g < tabs; ++g) {
            // (the tables of the block's g-th instance, or the shared ones)
            const long long ii = per_tab ? block_instance(A, g) : 0;
            const double *src = A.rowc + ii * A.table_stride;
            const int *srci = A.rowi + ii * A.tablei_stride;
            double *dst = lds_rowc + g * (NSUB * S::ROWC_L * GROUP);
            for (int s = 0; s < (ROWC_G ? 0 : NSUB); ++s)     // only the constants this shape's row evaluation reads
                for (int i = tid; i < S::ROWC_L * GROUP; i += nthreads) {
                    if constexpr (S::RCPAIR) {      // i = ((pair * 16 + row) * 2 + half)
                        const int c = (i / (2 * GROUP)) * 2 + (i & 1), row = (i >> 1) & (GROUP - 1);
                        dst[s * S::ROWC_L * GROUP + i] = c < UR_W1 - UR_SA + 1 ? src[(s * ROWC + S::RC0 + c) * GROUP + row] : 0.0;
                    } else {
                        dst[s * S::ROWC_L * GROUP + i] = src[(s * ROWC + S::RC0) * GROUP + i];
                    }
                }
            for (int i = tid; i < NSUB * ROWI * GROUP; i += nthreads) lds_rowi[g * (NSUB * ROWI * GROUP) + i] = srci[i];
        }
    }
    wv::block_sync();

    // this instance's image (a private one: only offsets >= IMG0 are there)
    const double *Mg = A.image + (per_inst && valid ? inst * A.image_stride : 0);   // ... and in HBM
    const double *M;
    if constexpr (LOW) M = Mg;
    else M = per_inst ? lds_img + (gib * S::IMGN - S::IMG0) : lds_img;
    const double *Ms = M + L.sub0;                                // current sub-problem block
    double *ubuf = lds_scr + gib * S::SCRATCH;
    double *ybuf = ubuf + S::UBUF;
    double *const meta_scr = ybuf + S::YBUF + RW_WORDS;      // the word after the report (Shape::RBUF = RW_WORDS + 1)
    static_assert(S::RBUF > RW_WORDS, "a spare word for the cache counters");

    // zero this instance's scratch once: with a padded shape (nu_io < NU) some u-tile entries
    // are read but never written
    for (int i = lig; i < S::SCRATCH; i += GROUP) ubuf[i] = 0.0;
    // ... and this wave's origin slabs: a linearisation that fails before anything was recorded
    // must leave a harmless (zero) extrapolation behind, not whatever the LDS held
    for (int i = lane; i < S::ORIGIN; i += 64) lds_scr[INST_PER_BLOCK * S::SCRATCH + wave * S::ORIGIN + i] = 0.0;
    wv::wave_fence();      // (the cache counters below land in the scratch that has just been zeroed)
    // the solution caches live in HBM between launches
    if (A.solver == SOLVER_CACHING_HOMOTOPY && valid)
        for (int s = 0; s < S::NSUBr; ++s)
            for (int i = lig; i < S::CACHEPM; i += GROUP) {
                const double v = A.cache[inst * S::CACHEIH + s * S::CACHE1H + i];
                if (i < NP * CACHE || !S::META_SCR) cache0[s * S::CACHE1 + i] = v;
                else if (i == NP * CACHE) meta_scr[0] = v;   // (count, head), two ints moved as one double
            }
    wv::wave_fence();

    // Which residual row (equation) this lane evaluates.  It starts as the host's row-order
    // hint and then follows the LU: whenever a factorisation has to interchange rows, the
    // lanes ADOPT the pivoted order, so the next factorisation finds its pivots in place.
    int rowid = lig;
    int grow_ = lig;     // ... and its row in the row-gathered copies (lanes beyond NN: the all-zero row, Layout::gs;
#define grow (*(L.gs == GROUP ? &rowid : &grow_))      /* the same number when the copies have 16 rows */
    RowDesc rd;
    const double *rc_norm = nullptr;
    auto load_rowdesc = [&]() ACME_LAMBDA {
        rd.kind = (lig < NN) ? rowi_s[0 * GROUP + rowid] : RK_NONE;
        rd.erow = rowi_s[1 * GROUP + rowid];
        rd.flags = rowi_s[2 * GROUP + rowid];
        rd.rc = rowc_s + (S::RCPAIR ? 2 : 1) * rowid;            // rc[c * GROUP] = row constant RC0 + c (RCPAIR: pair c at rc[c * 2 * GROUP])
        // condensed shapes: in the Newton loop the lanes of the linear rows evaluate the constants of row 15 --
        // res = e0, Jq = (1, 0, 0) (acme_pack.h) -- whatever row they hold
        if constexpr (COND) rc_norm = lig < NL ? rowc_s + 2 * (GROUP - 1) : rd.rc;
        if constexpr (L.gs != GROUP) grow = lig < NN ? rowid : NN;
        // register-cached constants: the kind-by-kind evaluation (RARE shapes) wants rc[0..7],
        // the unified rows sA sB cA cB dA dB h
        if constexpr (!S::RCPAIR)
            sfor<0, 8>([&](auto c_) ACME_LAMBDA { rd.k[decltype(c_)::value] = rd.rc[decltype(c_)::value * GROUP]; });
    };
    load_rowdesc();
    const bool has_bjt = A.has_bjt != 0;
    // The Newton loop's two wave-uniform parameters in VECTOR registers: as scalars they are part of an
    // 8-register kernel-argument tuple that the allocator spills and reloads whole -- 16 v_readlane per
    // Newton iteration to look at these two.
    const double tol_v = wv::keep(A.tol);
    const int maxiter_v = wv::keepi(A.maxiter);
    ExpTabV etv;
    if constexpr (S::EXPV) {
        const wv::ExpTab t0 = wv::load_exp_tab();
        sfor<0, 16>([&](auto ic) ACME_LAMBDA { etv.v[decltype(ic)::value] = wv::keep(t0[decltype(ic)::value]); });
    }

    // ---- persistent per-instance state, in registers for the whole launch --------------
    double x[NXSr];      // state vector, element s*16+lig
    double lp = 0.0;     // extrapolation origin: last_p[lig]
    double lz = 0.0;     //                       last_z[lig]
    // the origin's linearisation (elimination multipliers, 1/pivot, Jq non-zeros, pfull entries of
    // row lig) lives in LDS (ojp[slot * OS]): all the first-order extrapolated start
    // (src/solvers.jl:209-215) needs
    double z = 0.0;      // current iterate z[lig]
    double pf[NT];       // (q0 + pexp*p) at the q rows rd.tc[] of this lane's residual row
    // ---- condensed solve: state (see "condensed solve" below) ----------------------------------------------------
    // cd: the CONDENSATION for the potentiometer positions cpos -- lanes NL .. NN-1: the rows of
    // fq' = fq_N - fq_L W their residual row reads (term t, reduced unknown j); lanes 0 .. NL-1: cd[0] = their row of
    // W = A_LL^-1 A_LN, cd[1][0 .. NL-1] = their row of A_LL^-1.  In registers for the whole launch, rewritten only
    // when a potentiometer moves.
    double cd[3][NRr];
    double crw = 0.0;                 // lanes < NL: -r w of the row (res = v + crw i), for cpos
    double cpos = (double)NAN;        // lanes < NL: the position the condensation was made for (NaN: none yet)
    double pfr[NT];                   // pf' = pfull + fq_L zp_L at this lane's q rows (lanes < NL: pfr[0] = -zp_L)
    if constexpr (COND)
        sfor<0, 3>([&](auto tc_) ACME_LAMBDA { sfor<0, NR>([&](auto jc) ACME_LAMBDA { cd[decltype(tc_)::value][decltype(jc)::value] = 0.0; }); });
    sfor<0, NT>([&](auto tc_) ACME_LAMBDA { pfr[decltype(tc_)::value] = 0.0; });
    // per-row results of the latest evaluate!
    double a[NNr];       // J row -> LU row
    double res = 0.0;
    double tv[NT];

    const double *st = A.state + (valid ? inst : 0) * S::STATE;
    sfor<0, NXS>([&](auto sc) ACME_LAMBDA {
        constexpr int s = decltype(sc)::value;
        int i = s * GROUP + lig;
        x[s] = (valid && i < NX) ? st[i] : 0.0;
    });
    // saved context of every sub-problem: its extrapolation origin, its current lane->row
    // assignment and its latest solution; (lp, lz, rowid, ...) above are the live copies of
    // the sub-problem being solved
    double lps[NSUB], lzs[NSUB], zs[NSUB];
    int rowids[NSUB];
    int stales[NSUB];
    sfor<0, NSUB>([&](auto sc) ACME_LAMBDA {
        constexpr int s = decltype(sc)::value;
        lps[s] = (NP > 0 && valid && lig < NP) ? st[NX + s * NP + lig] : 0.0;
        lzs[s] = (NN > 0 && valid && lig < NN) ? st[NX + NSUB * NP + s * NN + lig] : 0.0;
        zs[s] = 0.0;
        stales[s] = 1;
        rowids[s] = valid ? A.roworder[(inst * NSUB + s) * GROUP + lig] : lig;
    });
    const int nsub = (NN > 0) ? A.nsub : 0;

    // ---- ACME_TIMING builds (tools/timing_probe.py): shader-clock time per code region ---
#ifdef ACME_TIMING
    enum { TB_POST, TB_PRE, TB_SETUP, TB_EVAL, TB_PIVOT, TB_GJ0, TB_GJP, TB_STORE, TB_GLUE, TB_HOMO,
           TB_E1, TB_E2, TB_E3, TB_N };
    long long tb[TB_N] = {0};
    long long tmark = (long long)__builtin_readcyclecounter();
#define ACME_T(bucket) do { wv::sched_fence(); long long t_ = (long long)__builtin_readcyclecounter(); \
                            tb[bucket] += t_ - tmark; tmark = t_; wv::sched_fence(); } while (0)
#ifdef ACME_TIMING_FINE
#define ACME_T2(bucket) ACME_T(bucket)
#else
#define ACME_T2(bucket) do { } while (0)
#endif
#else
#define ACME_T(bucket) do { } while (0)
#define ACME_T2(bucket) do { } while (0)
#endif

    // ---- helpers ------------------------------------------------------------------------
    // pfull <- q0 + pexp*p   (set_p closure, src/ACME.jl:237-243), only the entries this
    // lane's row needs
    // (FUSE: p_j reaches the lanes through the DPP operand of the multiply-add itself, see fmac_bcast)
    auto set_p = [&](double p) ACME_LAMBDA {
        // Batch first: the pexp rows (and q0) of all three terms are requested before the first multiply-add runs.
        // (Rounds 1-2 went load - wait - accumulate term by term behind a scheduling fence, to bound the registers
        // in flight: that exposed the LDS latency three times a sample, and registers are not scarce here,
        // before the solve.)
        // (pair layout with an odd np: the pad column of the last pair holds q0, see acme_pack.h)
        constexpr bool q0_in_pad = L.pairs && (NP % 2 == 1);
        double pe[NT][NPr + 1], q0v[NT];
        sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
            constexpr int t = decltype(tc_)::value;
            if constexpr (L.pairs) {
                sfor<0, (NP + 1) / 2>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = 2 * decltype(jc)::value;
                    const wv::pair_t v = wv::ld2(&Ms[L.pexpr + L.gat(t, j, 0, NP) + 2 * grow]);
                    pe[t][j] = v.lo;
                    pe[t][j + 1] = v.hi;
                });
            } else {
                sfor<0, NP>([&](auto jc) ACME_LAMBDA { pe[t][decltype(jc)::value] = Ms[L.pexpr + L.gat(t, decltype(jc)::value, 0, NP) + grow]; });
            }
            if constexpr (!q0_in_pad) q0v[t] = Ms[L.q0i(t, 0) + grow];
        });
        double pb[NPr];
        if constexpr (!S::FUSE) {
            sfor<0, NP>([&](auto jc) ACME_LAMBDA { pb[decltype(jc)::value] = wv::bcast16<decltype(jc)::value>(p); });
            wv::sched_fence();
        }
        sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
            constexpr int t = decltype(tc_)::value;
            double acc = q0_in_pad ? pe[t][NP] : q0v[t];
            if constexpr (S::FUSE) {       // one statement per term; the first one waits for p (DPP hazard)
                wv::fmac_bcast_chain<NP, true>(acc, p, pe[t]);
            } else {
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    acc = fma(pe[t][j], pb[j], acc);
                });
            }
            pf[t] = acc;
        });
    };

    // evaluate!(nleq, z): q = pfull + fq*z; (res, Jq) = elements(q); J = Jq*fq
    // (src/ACME.jl:178-188, src/circuit.jl:10-17).  Leaves J row in a[], residual in res,
    // the row's Jq non-zeros in tv/tc.  Returns true if res and J are finite.
    auto evaluate = [&](double zz) ACME_LAMBDA -> bool {
        // q[tc[t]] = pfull + fq*z for the (at most NT) q entries this row depends on: every
        // lane forms its own, so no cross-lane exchange of q is needed
        double zb[NNr];
        if constexpr (!S::FUSE)
            sfor<0, NN>([&](auto jc) ACME_LAMBDA { zb[decltype(jc)::value] = wv::bcast16<decltype(jc)::value>(zz); });
        // this row's fq entries (used twice: q = pf + fq*z here, J = Jq*fq below)
        double fqv[NT][NNr + 1];
        sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
            constexpr int t = decltype(tc_)::value;
            if constexpr (L.pairs) {
                sfor<0, (NN + 1) / 2>([&](auto jc) ACME_LAMBDA {      // two columns per LDS read (Layout::gat)
                    constexpr int j = 2 * decltype(jc)::value;
                    const wv::pair_t v = wv::ld2(&Ms[L.fqr + L.gat(t, j, 0, NN) + 2 * grow]);
                    fqv[t][j] = v.lo;
                    fqv[t][j + 1] = v.hi;
                });
            } else {
                sfor<0, NN>([&](auto jc) ACME_LAMBDA { fqv[t][decltype(jc)::value] = Ms[L.fqr + L.gat(t, decltype(jc)::value, 0, NN) + grow]; });
            }
        });
        // the row's 13 constants, staged in pairs (Shape::RCPAIR).  Small shapes request them HERE, with the fq
        // rows: their q chains are too short to cover a second LDS round trip started behind them
        double urc[14];
        auto load_urc = [&]() ACME_LAMBDA {
            sfor<0, 7>([&](auto pc) ACME_LAMBDA {
                constexpr int p = decltype(pc)::value;
                const wv::pair_t v = wv::ld2(&rd.rc[p * 2 * GROUP]);
                urc[2 * p] = v.lo;
                urc[2 * p + 1] = v.hi;
            });
        };
        constexpr bool URC_FIRST = !S::RARE && NN < 7;       // (nn = 7, config 4's shape: -0.5 % with it, 19 registers already spilled)
        if constexpr (URC_FIRST) load_urc();
        wv::sched_fence();
        double e[NT];
        if constexpr (!S::FUSE) {
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                double acc = pf[t];
                sfor<0, NN>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    acc = fma(fqv[t][j], zb[j], acc);
                });
                e[t] = acc;
            });
        } else {
            // z_j reaches the lanes through the DPP operand of the multiply-add; zz is the iterate
            // the caller updated a few instructions before
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA { e[decltype(tc_)::value] = pf[decltype(tc_)::value]; });
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {      // one statement per term; the first one waits for zz
                constexpr int t = decltype(tc_)::value;
                wv::fmac_bcast_chain<NN, true>(e[t], zz, fqv[t]);      // (every statement waits: the compiler may copy zz between them)
            });
        }
        ACME_T2(TB_E1);
        // hoisted exponentials: diode exp(v/(eta vT)), BJT exp(vE/..), exp(vC/..)
        double exA, exB;
        if constexpr (S::RARE) {
            const bool expo = rd.kind == RK_DIODE || rd.kind == RK_BJT;
            exA = exp_junction(expo ? e[0] * rd.k[0] : 0.0);
            exB = has_bjt ? exp_junction(rd.kind == RK_BJT ? e[1] * rd.k[1] : 0.0) : 1.0;
            ACME_T2(TB_E2);
            eval_row<true, NT>(rd, e, exA, exB, res, tv);
        } else {
            if constexpr (!URC_FIRST) load_urc();
            const double sA = urc[0], sB = urc[1];
            // (EXP2: no branch at all -- a row without a second junction has sB = 0, and exp_junction2 returns
            // exactly 1 for it, as the other path sets it; a model without any BJT in one of these shapes
            // pays ~20 wasted instructions per row evaluation, every other one saves a taken branch)
            if (S::EXP2 || ACME_USUAL(has_bjt)) {                     // sA/sB = 0: exp(0) = 1
                if constexpr (S::EXPV) exp_junction2(e[0] * sA, e[1] * sB, exA, exB, etv);
                else exp_junction2(e[0] * sA, e[1] * sB, exA, exB, wv::load_exp_tab());     // (the Jacobian export of the condensed shapes)
            } else {
                // (big shape: this path -- models without a BJT -- keeps the scalar table; on the register
                // table it costs the two-exponential path 4 spilled registers and 8 % of its speed)
                if constexpr (!S::MULT && S::EXPV) exA = exp_junction(e[0] * sA, etv);
                else exA = exp_junction(e[0] * sA);
                exB = 1.0;
            }
            ACME_T2(TB_E2);
            eval_row_unified_c<NT>(urc, e, exA, exB, res, tv);
        }
        ACME_T2(TB_E3);
        sfor<0, NN>([&](auto jc) ACME_LAMBDA {   // J row = Jq row * fq (src/ACME.jl:186)
            constexpr int j = decltype(jc)::value;
            double acc = tv[0] * fqv[0][j];
            sfor<1, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                acc = fma(tv[t], fqv[t][j], acc);
            });
            a[j] = acc;
        });
        if constexpr (L.pairs && NN % 2 == 1) {
            // the pad halves of the rows' last pairs, "used" here: dead on arrival, their registers were handed
            // to the NEXT load of the batch -- which then had to wait for the whole batch before it could issue
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA { wv::touch(fqv[decltype(tc_)::value][NN]); });
        }
        // Non-finite anywhere in this instance's res / J (src/solvers.jl:219-221: solve() returns at once)?  No
        // test of its own here: a NaN in res or J turns the elimination's result into NaN, an infinite residual
        // too (inf * 0), an infinite entry of J ends up as a pivot (the pivot search prefers it) whose stored
        // reciprocal is 0 -- solve_inplace reports all three (`mine`), and the caller stops exactly like the
        // reference does: hasconverged = (max |res| < tol), which a non-finite residual fails.  (A separate
        // finiteness ballot cost 9 vector instructions per evaluate!.)
        return true;
    };

    // calc_Jp closure (src/ACME.jl:246-251): Jp row = Jq row * pexp
    auto calc_jp = [&](double (&jp)[NPr]) ACME_LAMBDA {
        if constexpr (L.pairs) {
            sfor<0, (NP + 1) / 2>([&](auto jc) ACME_LAMBDA {
                constexpr int j = 2 * decltype(jc)::value;
                const wv::pair_t v0 = wv::ld2(&Ms[L.pexpr + L.gat(0, j, 0, NP) + 2 * grow]);
                double a0 = tv[0] * v0.lo, a1 = tv[0] * v0.hi;
                sfor<1, NT>([&](auto tc_) ACME_LAMBDA {
                    constexpr int t = decltype(tc_)::value;
                    const wv::pair_t v = wv::ld2(&Ms[L.pexpr + L.gat(t, j, 0, NP) + 2 * grow]);
                    a0 = fma(tv[t], v.lo, a0);
                    a1 = fma(tv[t], v.hi, a1);
                });
                jp[j] = a0;
                if constexpr (j + 1 < NP) jp[j + 1] = a1;
            });
        } else {
            double pv[NT][NPr];      // (all entries requested first: column by column, every column waited for its own reads)
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                sfor<0, NP>([&](auto jc) ACME_LAMBDA { pv[t][decltype(jc)::value] = Ms[L.pexpr + L.gat(t, decltype(jc)::value, 0, NP) + grow]; });
            });
            wv::sched_fence();
            sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                double acc = tv[0] * pv[0][j];
                sfor<1, NT>([&](auto tc_) ACME_LAMBDA {
                    constexpr int t = decltype(tc_)::value;
                    acc = fma(tv[t], pv[t][j], acc);
                });
                jp[j] = acc;
            });
        }
    };

    // After a pivoted factorisation the lane at position i holds what was row orig[i]: make
    // that the lane's row from now on (row descriptor and the latest Jq non-zeros move along).
    auto adopt = [&](int orig) ACME_LAMBDA {
        rowid = wv::shfl16(rowid, orig);
        sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
            constexpr int t = decltype(tc_)::value;
            tv[t] = wv::shfl16(tv[t], orig);
            pf[t] = wv::shfl16(pf[t], orig);
        });
        if constexpr (COND) {       // the row's reduced pfull entries and its rows of the condensation move along
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                pfr[decltype(tc_)::value] = wv::shfl16(pfr[decltype(tc_)::value], orig);
            });
            sfor<0, 3>([&](auto tc_) ACME_LAMBDA {
                sfor<0, NR>([&](auto jc) ACME_LAMBDA { cd[decltype(tc_)::value][decltype(jc)::value] = wv::shfl16(cd[decltype(tc_)::value][decltype(jc)::value], orig); });
            });
            crw = wv::shfl16(crw, orig);
            cpos = wv::shfl16(cpos, orig);
        }
        load_rowdesc();
    };

    int stale = 1;       // (an integer in a vector register, like the loop flags of base_solve; see linearize)
    // ======================= condensed solve (Shape::NL > 0) ===========================================================
    // The residual rows of a potentiometer,  res = v - r w i  with w = pos or 1 - pos  (src/elements.jl:25-30), are LINEAR
    // in z for a given p when `pos` comes from the inputs alone (its fq row is zero, src/ACME.jl:176-189):
    //     A_L(pos) z + b_L(p) = 0,    A_L = fq[v] - r w fq[i],    b_L = pfull[v] - r w pfull[i].
    // The reference's Newton iteration (src/solvers.jl:207-236) solves these rows exactly in its first step and keeps
    // them solved: all it iterates on are the other NR = NN - NL rows.  With the linear rows in lanes 0 .. NL-1 and
    // their pivot columns first (acme_pack.h: CondPlan), z = (z_L, z_N):
    //     z_L = zp_L - W z_N,   W = A_LL^-1 A_LN,   zp_L = -A_LL^-1 b_L
    //     q = pfull + fq z = pf' + fq' z_N,   pf' = pfull + fq_L zp_L,   fq' = fq_N - fq_L W        (condensation)
    // and the Jacobian of the reduced system,  S = Jq_N fq',  is the Schur complement the full elimination reaches
    // after its first NL steps when it takes its pivots from the linear rows -- so the Newton loop forms S directly
    // (NT x NR multiply-adds instead of NT x NN, twice) and eliminates NR steps instead of NN; the lanes of the linear
    // rows ride along with their row of W (they evaluate the constants of row 15: res = e0 = -(zp_L - W z_N),
    // Jq = (1, 0, 0)), so that the same elimination hands them -z_L of the new iterate: z_L never needs a pass of
    // its own.  W, A_LL^-1 and fq' depend on the potentiometer positions only: `condense` recomputes them (Gauss-Jordan
    // of the linear rows with the others riding along) when a position changes -- never, in a sweep with fixed pots --
    // and they live in registers (cd) in between.  Same Newton iterates as the reference's, to rounding
    // (tools/condense_proto.py compares the algebra with the oracle: identical iteration counts, outputs to 1e-14).
    // An iterate OFF the subspace of the linear rows -- the extrapolated start when the origin was not on it: initial
    // solution, acme_batch_set_state, or after a potentiometer moved -- is evaluated on the full q (`off`), its linear
    // residuals take part in the convergence test, and the step lands on the subspace as the reference's does:
    //     S dz_N = F_N - Jq_N (q - q'),   q' = pf' + fq' z_N.
    const bool islin = COND && lig < NL;
    const double sgn = islin ? 0.0 : 1.0;          // z <- sgn z - dz: the lanes of the linear rows get -dz = the new z_L
    int osub = 1;                                  // the origin (lp, lz) may lie off the subspace (KArgs::cflags)
    if constexpr (COND) osub = valid ? (A.cflags[inst] & 1) : 0;
    double opos = (double)NAN;                     // lanes < NL: the potentiometer position at the origin (lp, lz)
    double corr = 0.0, lres = 0.0, efl1 = 0.0;
    auto inst_any = [&](bool x) ACME_LAMBDA -> bool { return ((wv::ballot(x) >> (grp * GROUP)) & 0xFFFFull) != 0ull; };
    // the rows' fq entries (two columns per LDS read), as in evaluate
    auto load_fq_rows = [&](double (&fqv)[NT][NNr + 1]) ACME_LAMBDA {
        sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
            constexpr int t = decltype(tc_)::value;
            sfor<0, (NN + 1) / 2>([&](auto jc) ACME_LAMBDA {
                constexpr int j = 2 * decltype(jc)::value;
                const wv::pair_t v = wv::ld2(&Ms[L.fqr + L.gat(t, j, 0, NN) + 2 * grow]);
                fqv[t][j] = v.lo;
                fqv[t][j + 1] = v.hi;
            });
        });
    };
    // W, A_LL^-1 and fq' for the potentiometer positions of the pfull entries set_p has just formed, for the instances
    // with `upd`.  The linear rows keep the order the lanes have adopted for them unless a pivot fails the threshold
    // (then: the reference's partial pivoting among them, once).
    auto condense = [&](bool upd) ACME_LAMBDA {
        if constexpr (COND) {
            // (one term's rows at a time, the results selected into cd as they come: the whole of it at once needed
            // ~180 registers on top of what the kernel holds, and the allocator paid for that on the usual path)
            auto load_fq_row = [&](auto tc_, auto &row) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                sfor<0, (NN + 1) / 2>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = 2 * decltype(jc)::value;
                    const wv::pair_t v = wv::ld2(&Ms[L.fqr + L.gat(t, j, 0, NN) + 2 * grow]);
                    row[j] = v.lo;
                    row[j + 1] = v.hi;
                });
            };
            double al[NNr], c6[NLr], hw = 0.0;
            int tries = 0;
            for (;;) {
                tries = wv::opaque(tries);
                // A_L row = Jq row * fq of a potentiometer half: (g0, h w, .) (acme_common.h: UnifiedRowConst)
                const wv::pair_t h_ = wv::ld2(&rd.rc[3 * 2 * GROUP]), g01 = wv::ld2(&rd.rc[4 * 2 * GROUP]),
                                 g2w0 = wv::ld2(&rd.rc[5 * 2 * GROUP]), w1_ = wv::ld2(&rd.rc[6 * 2 * GROUP]);
                hw = h_.lo * fma(w1_.lo, pf[2], g2w0.hi);
                {
                    double r0[NNr + 1], r1[NNr + 1];
                    load_fq_row(std::integral_constant<int, 0>{}, r0);
                    load_fq_row(std::integral_constant<int, 1>{}, r1);
                    sfor<0, NN>([&](auto jc) ACME_LAMBDA {
                        constexpr int j = decltype(jc)::value;
                        al[j] = fma(hw, r1[j], g01.lo * r0[j]);
                    });
                }
                // A_LL^-1 (rows in the lanes): Gauss-Jordan of [A_LL | I]
                double a6[NLr], b6 = 0.0, d6;
                sfor<0, NL>([&](auto kc) ACME_LAMBDA {
                    constexpr int k = decltype(kc)::value;
                    a6[k] = al[k];
                    c6[k] = lig_eq<k>() ? 1.0 : 0.0;
                });
                unsigned long long viol = LUL::template solve_inplace<NL, false, S, true, true, true>(a6, b6, c6, nullptr, false, d6);
                viol &= wv::ballot(islin && upd);
                if (ACME_USUAL(viol == 0ull || tries != 0)) break;
                const bool mine = ((viol >> (grp * GROUP)) & 0xFFFFull) != 0ull;
                sfor<0, NL>([&](auto kc) ACME_LAMBDA { a6[decltype(kc)::value] = al[decltype(kc)::value]; });
                int orig;
                (void)LUL::template pivot_order_range<0, NL>(a6, orig, lig, grp);
                orig = (mine && islin) ? orig : lig;
                adopt(orig);
                stale = mine ? 1 : stale;       // the slab's entries of these lanes belong to the rows they held
                tries = 1;
            }
            // W (lanes < NL) and fq' (the others, one pass per term of their rows): the elimination of the linear rows
            // with every row's columns NL .. NN-1 riding along
            sfor<0, 3>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                double a_[NNr + 1], b_ = 0.0, none[1] = {0.0}, dv;
                load_fq_row(tc_, a_);
                sfor<0, NN>([&](auto jc) ACME_LAMBDA { a_[decltype(jc)::value] = islin ? al[decltype(jc)::value] : a_[decltype(jc)::value]; });
                double ar[NNr];
                sfor<0, NN>([&](auto jc) ACME_LAMBDA { ar[decltype(jc)::value] = a_[decltype(jc)::value]; });
                (void)LU::template solve_range<0, NL, 0, false, S, true>(ar, b_, none, nullptr, false, dv);
                sfor<0, NR>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    const double nv = t == 0 ? (islin ? ar[NL + j] * dv : ar[NL + j])
                                             : (islin ? (t == 1 && j < NL ? c6[j < NL ? j : 0] : 0.0) : ar[NL + j]);
                    cd[t][j] = upd ? nv : cd[t][j];
                });
            });
            crw = upd ? hw : crw;
            cpos = upd ? pf[2] : cpos;
            ACME_T(TB_GJ0);     // (the timing builds' "GJ<0>" bucket: unused by the condensed kernel's Newton loop)
        }
    };
    // pf (set_p) -> pf' for the current condensation; lanes < NL: pfr[0] = -zp_L = A_LL^-1 b_L
    auto prep_reduced = [&]() ACME_LAMBDA {
        if constexpr (COND) {
            double fql[NT][NLr + 1];       // fq_L entries of this lane's rows: the first NL columns of the row-gathered copies
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                sfor<0, (NL + 1) / 2>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = 2 * decltype(jc)::value;
                    const wv::pair_t v = wv::ld2(&Ms[L.fqr + L.gat(t, j, 0, NN) + 2 * grow]);
                    fql[t][j] = v.lo;
                    fql[t][j + 1] = v.hi;
                });
            });
            const double bl = fma(crw, pf[1], pf[0]);
            double nzp = 0.0;
            wv::fmac_bcast_chain<NL, true>(nzp, bl, cd[1]);
            const double zpl = -nzp;
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                pfr[t] = pf[t];
                wv::fmac_bcast_chain<NL, true>(pfr[t], zpl, fql[t]);      // (every statement waits: they are not ordered among themselves)
            });
            pfr[0] = islin ? nzp : pfr[0];
        }
    };
    // evaluate! on the reduced system: q' = pf' + fq' z_N, (res, Jq) = elements(q'), S row = Jq row * fq'.  Leaves the S row
    // in a[NL ..], the residual in res, the right-hand side of the elimination in rhs.  off (offany: any instance of the
    // wave): this iterate may lie off the linear rows' subspace -- the elements see the full q = pfull + fq z, the step
    // corrects for the difference, and lres is the lane's linear residual (lanes < NL).
    auto evaluate_c = [&](double zz, bool offany, bool off) ACME_LAMBDA {
        if constexpr (COND) {
            double urc[14];
            sfor<0, 7>([&](auto pc) ACME_LAMBDA {
                constexpr int p = decltype(pc)::value;
                const wv::pair_t v = wv::ld2(&rc_norm[p * 2 * GROUP]);
                urc[2 * p] = v.lo;
                urc[2 * p + 1] = v.hi;
            });
            double e[NT], de[NT];       // de: full q minus reduced q' (off only)
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA { e[decltype(tc_)::value] = pfr[decltype(tc_)::value]; });
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int t = decltype(tc_)::value;
                wv::fmac_bcast_chain_from<NL, NR, t == 0>(e[t], zz, cd[t]);
            });
            if (ACME_RARE(offany)) {
                double fqv[NT][NNr + 1], ef[NT];
                load_fq_rows(fqv);
                sfor<0, NT>([&](auto tc_) ACME_LAMBDA { ef[decltype(tc_)::value] = pf[decltype(tc_)::value]; });
                sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                    constexpr int t = decltype(tc_)::value;
                    wv::fmac_bcast_chain<NN, true>(ef[t], zz, fqv[t]);
                });
                lres = fma(crw, ef[1], ef[0]);
                efl1 = ef[1];
                sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                    constexpr int t = decltype(tc_)::value;
                    de[t] = ef[t] - e[t];
                    e[t] = (off && !islin) ? ef[t] : e[t];
                });
            }
            double exA, exB;
            if constexpr (S::EXPV) exp_junction2(e[0] * urc[0], e[1] * urc[1], exA, exB, etv);
            else exp_junction2(e[0] * urc[0], e[1] * urc[1], exA, exB, wv::load_exp_tab());
            eval_row_unified_c<NT>(urc, e, exA, exB, res, tv);
            sfor<0, NR>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                double acc = tv[0] * cd[0][j];
                sfor<1, NT>([&](auto tc_) ACME_LAMBDA { acc = fma(tv[decltype(tc_)::value], cd[decltype(tc_)::value][j], acc); });
                a[NL + j] = acc;
            });
            if (ACME_RARE(offany)) {
                double c_ = 0.0;
                sfor<0, NT>([&](auto tc_) ACME_LAMBDA { c_ = fma(tv[decltype(tc_)::value], de[decltype(tc_)::value], c_); });
                corr = (off && !islin) ? c_ : 0.0;
            }
        }
    };

    // One pass of the condensed Newton iteration: evaluate_c, then the NR steps of the elimination with the residual
    // riding along (dz); for an iterate whose residual is below tol (or `force`) the lanes record the steps, the row's
    // Jq non-zeros and its pf' entries in the origin slab, as linearize does.  Returns `mine`: this instance's pivots
    // failed the threshold, or the result is not finite -- the caller leaves the Newton loop, lets the lanes re-learn
    // their order (relearn_c) and repeats the pass.  Nothing in here assigns what the Newton loop only reads (cd, pf',
    // the lanes' rows): kept inside the loop, the re-learning made them loop-carried values with one register copy
    // each -- 30 of them -- per iteration.
    auto lin_c = [&](double zz, bool act, unsigned long long actm, bool force, bool &small, double &dz, bool offany, bool off) ACME_LAMBDA -> bool {
        bool mine = false;
        if constexpr (COND) {
            evaluate_c(zz, offany, off);
            ACME_T(TB_EVAL);
            unsigned long long big = wv::ballot(!(fabs(res) < tol_v)) & rows4(((1ull << NN) - 1ull) & ~((1ull << NL) - 1ull));
            if (ACME_RARE(offany)) big |= wv::ballot(off && islin && !(fabs(lres) < tol_v));     // (off the subspace, the linear rows count)
            small = ((big >> (grp * GROUP)) & 0xFFFFull) == 0ull;
            const bool want = force || (act && small);
            dz = res;
            if (ACME_RARE(offany)) dz = res - corr;
            double none[1] = {0.0}, dinv = 0.0;
            const bool recording = wv::ballot(want) != 0ull;
            unsigned long long viol = LU::template solve_range<NL, NN, 0, true, S>(a, dz, none, ojp, want && lig < NN, dinv);
            viol &= actm | wv::ballot(force);
            mine = ((viol >> (grp * GROUP)) & 0xFFFFull) != 0ull;
            ACME_T(TB_GJP);
            if (recording) {
                if (want && !mine && lig < NN)
                    sfor<(NR + 1) / 2, S::OSLOTS / 2>([&](auto cc) ACME_LAMBDA {
                        constexpr int c = 2 * decltype(cc)::value;
                        auto slotv = [&](auto sc) ACME_LAMBDA -> double {
                            constexpr int sl = decltype(sc)::value;
                            if constexpr (sl < S::OS_TV + NT) return tv[sl - S::OS_TV];
                            else if constexpr (sl < S::OS_PF + NT) return pfr[sl - S::OS_PF];
                            else return 0.0;
                        };
                        wv::st2(&ojp[S::oslot(c)], slotv(std::integral_constant<int, c>{}), slotv(std::integral_constant<int, c + 1>{}));
                    });
                stale = want ? (mine ? 1 : 0) : stale;
            }
            ACME_T(TB_STORE);
        }
        return mine;
    };
    // the instances with `who` re-learn the order of their reduced system's rows: the reference's partially pivoted LU
    // (src/solvers.jl:58-78) on S at zz, just to find the pivot order, which the lanes adopt.  false: S is singular.
    auto relearn_c = [&](double zz, bool who, bool offany, bool off) ACME_LAMBDA -> bool {
        bool okp = true;
        if constexpr (COND) {
            evaluate_c(zz, offany, off);
            int orig;
            okp = LU::template pivot_order_range<NL, NN>(a, orig, lig, grp);
            orig = who ? orig : lig;
            adopt(orig);
            stale = who ? 1 : stale;
            okp = who ? okp : true;
            ACME_T(TB_PIVOT);
        }
        return okp;
    };

    // One Newton linearisation at z: evaluate! (res, J), then solve J dz = res by in-place
    // Gauss-Jordan.  An iterate whose residual is already below tol is going to be accepted and
    // become the new extrapolation origin (hasconverged looks at the residual alone,
    // src/solvers.jl:203,225-233); for it -- `want`, or `force` -- the instance's lanes record the
    // elimination (their multipliers and 1/pivot), the row's Jq non-zeros and its pfull entries in
    // the origin slab (set_extrapolation_origin, src/solvers.jl:191-196: the reference keeps the
    // factors and Jp there; Jp (p' - p) = Jq (pexp p' - pexp p) needs only Jq and pfull).  If the
    // in-place pivots were not the maxima (a few % of the calls) a rare, out-of-line block evaluates
    // again, runs the reference's partially pivoted LU to learn the pivot order, lets the lanes adopt
    // it, evaluates in the new order and eliminates again.  (Rounds 1-2 ran the three stages as one
    // loop over an opaque `phase` so that evaluate / pivot_order had ONE call site each; the usual
    // pass as straight-line code saves the loop head's register copies and two branches per Newton
    // iteration.)  finite: res and J finite; ok: J non-singular; small: |res| < tol.
    // `stale`: the slab no longer describes the origin (lp, lz) in the lanes' current order -- an
    // instance changed its row order without storing a new origin, or a recorded elimination was
    // discarded -- and has to be rebuilt before the next extrapolation (cached_solve does).
    // (actm: the caller's ballot of `act` -- it has it anyway, as its loop condition)
    auto linearize = [&](double zz, bool act, unsigned long long actm, bool force, bool &finite, bool &ok, bool &small, double &dz) ACME_LAMBDA {
        int okf = 1;     // `ok`: carried as an integer in a vector register
        bool want, recording, mine;
        double jp[NPr];
        double dinv = 0.0;       // (MULT, NN even: the slot left over by the recorded elimination's pairs)
        // the elimination on the latest evaluate!; returns the lanes that tripped the pivot threshold (or got a
        // non-finite result), restricted to the instances whose result is used
        auto eliminate = [&]() ACME_LAMBDA -> unsigned long long {
            // only the boolean is needed, so no max-reduction -- one compare and a ballot; a NaN
            // residual counts as not small
            const unsigned long long big = wv::ballot(!(fabs(res) < tol_v)) & rows4((1ull << NN) - 1ull);
            small = ((big >> (grp * GROUP)) & 0xFFFFull) == 0ull;
            want = force || (act && finite && small);
            unsigned long long viol;
            double none[1] = {0.0};
            dz = res;
            recording = wv::ballot(want) != 0ull;
            if constexpr (S::MULT) {
                // ONE elimination for iterates that become the origin and for those that do not: the recording
                // costs no arithmetic (the multipliers exist anyway), only the predicated stores at its end
                viol = LU::template solve_inplace<0, true, S, S::GJHEAD, S::SAFE0>(a, dz, none, ojp, S::LITROWS ? and_rows<(1ull << NN) - 1ull, true>(want) : (want && lig < NN), dinv);
                ACME_T(TB_GJP);
            } else if (recording) {       // the columns of Jp ride along: jp <- J^-1 Jp
                calc_jp(jp);
                viol = LU::template solve_inplace<NP, false, S, S::GJHEAD, S::SAFE0>(a, dz, jp, ojp, false, dinv);
                ACME_T(TB_GJP);
            } else {
                viol = LU::template solve_inplace<0, false, S, S::GJHEAD, S::SAFE0>(a, dz, none, ojp, false, dinv);
                ACME_T(TB_GJ0);
            }
            // the other instances' results are not used.  (The caller's own ballot of `act`, or a fresh one: the
            // birdie's lone waves lose 3.5 % with the mask kept alive across the iteration, the superover shapes
            // gain 0.4 ... 0.8 %)
            if constexpr (S::ACTM) viol &= actm | wv::ballot(force);
            else viol &= wv::ballot(act || force);
            mine = ((viol >> (grp * GROUP)) & 0xFFFFull) != 0ull;
            return viol;
        };
        // The usual pass, as straight-line code (no phase variable, no loop head to copy registers at) ...
        finite = evaluate(zz);
        ACME_T(TB_EVAL);
        if (ACME_RARE(eliminate() != 0ull)) {
            // ... and the rare one: some instance of the wave has to re-learn its pivot order.  Only the
            // instances that tripped the threshold change their row order: what an instance computes must
            // not depend on which other instances share its wave.
            const bool relearn = mine;
            (void)evaluate(zz);                  // J once more (the elimination has consumed it)
            int orig;
            const bool okp = LU::pivot_order(a, orig, lig, grp);
            okf = relearn ? (okp ? 1 : 0) : 1;
            orig = relearn ? orig : lig;
            adopt(orig);
            if constexpr (S::MULT) stale = relearn ? 1 : stale;   // the recorded elimination is per row order
            ACME_T(TB_PIVOT);
            finite = evaluate(zz);               // ... in the new row order
            (void)eliminate();
        }
        okf = mine ? 0 : okf;
        if (recording) {
            if (S::LITROWS ? and_rows<(1ull << NN) - 1ull, true>(want && !mine) : (want && !mine && lig < NN)) {   // per-lane predicated LDS stores
                if constexpr (S::MULT) {
                    // the rest of the entry -- the row's Jq non-zeros and pfull entries -- as 16-byte pairs
                    // too (the multipliers and 1/pivot went in at the end of the elimination)
                    sfor<(NN + 1) / 2, S::OSLOTS / 2>([&](auto cc) ACME_LAMBDA {
                        constexpr int c = 2 * decltype(cc)::value;
                        auto slotv = [&](auto sc) ACME_LAMBDA -> double {
                            constexpr int sl = decltype(sc)::value;
                            if constexpr (sl == NN) return dinv;
                            else if constexpr (sl < S::OS_TV + NT) return tv[sl - S::OS_TV];
                            else if constexpr (sl < S::OS_PF + NT) return pf[sl - S::OS_PF];
                            else return 0.0;
                        };
                        wv::st2(&ojp[S::oslot(c)], slotv(std::integral_constant<int, c>{}),
                                slotv(std::integral_constant<int, c + 1>{}));
                    });
                } else {
                    sfor<0, NP>([&](auto jc) ACME_LAMBDA { ojp[decltype(jc)::value * OS] = jp[decltype(jc)::value]; });
                }
            }
            // MULT: recorded and kept -> the slab is the origin-to-be; recorded but unusable -> it
            // is nothing.  Otherwise the slab is only written when the result is kept.
            if constexpr (S::MULT) stale = want ? (mine ? 1 : 0) : stale;
            else stale = (want && !mine) ? 0 : stale;
        }
        ACME_T(TB_STORE);
        ok = okf != 0;
    };

    // switch the live solver context to sub-problem s / save it back
    auto enter_sub = [&](auto sc) ACME_LAMBDA {
        constexpr int s = decltype(sc)::value;
        Ms = M + L.sub0 + s * L.sub_stride;
        rowc_s = rowc_b + s * (ROWC_G ? ROWC * GROUP : S::ROWC_L * GROUP);
        rowi_s = rowi_b + s * ROWI * GROUP;
        ojp = ojp0 + s * S::ORIGIN1;
        cch = cache0 + s * S::CACHE1;
        czg = A.cache + (valid ? inst : 0) * S::CACHEIH + s * S::CACHE1H + S::CACHEPM;
        lp = lps[s];
        lz = lzs[s];
        rowid = rowids[s];
        stale = stales[s];
        load_rowdesc();
    };
    auto leave_sub = [&](auto sc) ACME_LAMBDA {
        constexpr int s = decltype(sc)::value;
        lps[s] = lp;
        lzs[s] = lz;
        rowids[s] = rowid;
        stales[s] = stale;
    };

    // set_extrapolation_origin(solver, p, z) (src/solvers.jl:183-196): the factors and Jp at the
    // origin are recomputed from (p, z), so only (p, z) has to persist in HBM.  That happens
    // lazily, before the first base solve of each sub-problem (`stale` starts true), through the same
    // code as a change of origin on a solution-cache hit (cached_solve).
    if (S::NSUB == 1) enter_sub(std::integral_constant<int, 0>{});   // stays entered for the whole launch

    // solve(::SimpleSolver, p) (src/solvers.jl:207-236) for the instances with `need`;
    // returns hasconverged, leaves needediterations in `its`.
    auto base_solve = [&](double target, bool need, int &its) ACME_LAMBDA -> bool {
        double mul[NN + 2], oj[NPr];
        if constexpr (S::MULT) LU::template load_stored<S>(mul, ojp);     // requested first: needed last, ~60 instructions on
        else if constexpr (NN < 7) sfor<0, NP>([&](auto jc) ACME_LAMBDA { oj[decltype(jc)::value] = ojp[decltype(jc)::value * OS]; });   // (likewise; nn = 7: no gain)
        set_p(target);
        // z <- last_z - last_J \\ (last_Jp * (p - last_p))  (src/solvers.jl:209-215).  Row r of
        // last_Jp (p - last_p) is  sum_t Jq[r, tc_t] (pfull(p) - pfull(last_p))[tc_t]  (Jp = Jq pexp,
        // src/ACME.jl:246-251): the origin's Jq non-zeros times the change of this row's pfull
        // entries, which set_p has just formed; then the origin's recorded elimination.
        double t = 0.0;
        if constexpr (S::MULT) {
            double otp[2 * NT + 2];                 // the origin's Jq non-zeros and pfull entries, two slots per LDS read
            constexpr int s0 = S::OS_TV & ~1;
            sfor<0, (S::OS_PF + NT - s0 + 1) / 2>([&](auto cc) ACME_LAMBDA {
                constexpr int c = 2 * decltype(cc)::value;
                const wv::pair_t v = wv::ld2(&ojp[S::oslot(s0 + c)]);
                otp[c] = v.lo;
                otp[c + 1] = v.hi;
            });
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int tt = decltype(tc_)::value;
                t = fma(otp[S::OS_TV - s0 + tt], pf[tt] - otp[S::OS_PF - s0 + tt], t);
            });
            LU::apply_loaded(t, mul);
        } else {       // the slab holds J^-1 Jp, row lig
            const double dp = target - lp;
            sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                t = fma(NN < 7 ? oj[j] : ojp[j * OS], wv::bcast16<j>(dp), t);
            });
        }
        z = sel(need, lz - t, z);
        // the loop-carried per-lane flags as integers in vector registers (bit 0 act, 1 conv, 2 accepted):
        // as 64-bit lane masks they were spilled and re-read through v_writelane / v_readlane every pass
        int fl = wv::keepi(need ? 1 : 0);
        its = 0;
        ACME_T(TB_SETUP);
        unsigned long long actm = wv::ballot((fl & 1) != 0);
        do {
            const bool act = (fl & 1) != 0;
            its += fl & 1;
            bool finite, ok, small;
            double dz;
            linearize(z, act, actm, false, finite, ok, small, dz);
            ACME_DBG("newton it %d lane %d act %d z %.17g res %.17g dz %.17g finite %d ok %d small %d", its, lane, (int)act, z, res, dz, (int)finite, (int)ok, (int)small);
            const bool want = act && finite && ok && small;
            const bool stop_bad = act && (!finite || !ok);
            const bool step = act && !stop_bad && !want;
            z = sel(step, z - dz, z);
            int nf = fl & ~1;
            // hasconverged is `resmaxabs < tol` even when solve() returned early because J was non-finite
            // or singular (src/solvers.jl:203,219-224); `small` is false for a NaN / inf residual
            nf = stop_bad ? (small ? (nf | 2) : (nf & ~2)) : nf;
            nf = want ? (nf | 4) : nf;
            nf = (step && its < maxiter_v) ? (nf | 1) : nf;
            fl = wv::keepi(nf);
            ACME_T(TB_GLUE);
        } while ((actm = wv::ballot((fl & 1) != 0)) != 0ull);
        const bool accepted = (fl & 4) != 0;
        lz = sel(accepted, z, lz);
        lp = sel(accepted, target, lp);
        return (fl & 6) != 0;
    };

    // solve(::SimpleSolver, p) on the condensed system.  cached_solve has run set_p(target), brought the condensation up
    // to date, formed pf' and requested the origin's recorded multipliers (cmul).  mv: the potentiometers moved since
    // the origin was taken -- the start is z0m (the full system's first-order extrapolation, cached_solve) and, like a
    // start from an origin that was itself off the linear rows' subspace (osub), is evaluated `off` it.
    // Flags (one integer per lane): 1 active, 2 hasconverged, 4 accepted, 8 this pass is `off`, 16 accepted off the
    // subspace, 32 the lanes have re-learnt their order for this pass, 64 the last residual test.
    auto base_solve_c = [&](double target, bool need, int &its, const double (&cmul)[NRr + 1], bool mv, double z0m) ACME_LAMBDA -> bool {
        int fl = 0;
        if constexpr (COND) {
            // z <- last_z - last_J \\ (last_Jp (p - last_p)) (src/solvers.jl:209-215) on the reduced system: the origin's Jq
            // non-zeros times the change of pf', then its recorded elimination; the lanes of the linear rows hold
            // (1, 0, 0) and -zp_L there and ride along: the start's z_L comes out of the same replay
            double otp[2 * NT + 2];
            constexpr int s0 = S::OS_TV & ~1;
            sfor<0, (S::OS_PF + NT - s0 + 1) / 2>([&](auto cc) ACME_LAMBDA {
                constexpr int c = 2 * decltype(cc)::value;
                const wv::pair_t v = wv::ld2(&ojp[S::oslot(s0 + c)]);
                otp[c] = v.lo;
                otp[c + 1] = v.hi;
            });
            double t = 0.0;
            sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                constexpr int tt = decltype(tc_)::value;
                t = fma(otp[S::OS_TV - s0 + tt], pfr[tt] - otp[S::OS_PF - s0 + tt], t);
            });
            LU::template apply_loaded_from<NL, NR>(t, cmul);
            z = sel(need, lz - t, z);
            z = sel(need && mv, z0m, z);
            fl = wv::keepi((need ? 1 : 0) | ((need && (osub != 0 || mv)) ? 8 : 0));
            its = 0;
            ACME_T(TB_SETUP);
            for (;;) {
                unsigned long long actm = wv::ballot((fl & 1) != 0), redo = 0ull;
                bool retry = false;
                do {
                    const bool act = (fl & 1) != 0, off = (fl & 8) != 0;
                    bool small;
                    double dz;
                    const bool mine = lin_c(z, act, actm, false, small, dz, wv::ballot(off) != 0ull, off);
                    retry = act && mine && (fl & 32) == 0;                 // first failure of the pivots in place: re-learn, repeat
                    const bool stop_bad = act && mine && !retry;           // (src/solvers.jl:219-224: J singular or not finite)
                    const bool want = act && !mine && small;
                    const bool step = act && !mine && !want;
                    its += (act && !retry) ? 1 : 0;
                    z = sel(step, fma(sgn, z, -dz), z);
                    int nf = fl;
                    nf = stop_bad ? ((nf & ~3) | (small ? 2 : 0)) : nf;      // hasconverged is the residual test alone (:203)
                    nf = want ? ((nf & ~1) | 4 | ((nf & 8) ? 16 : 0)) : nf;
                    nf = (step && !(its < maxiter_v)) ? (nf & ~1) : nf;
                    nf = retry ? (small ? (nf | 64) : (nf & ~64)) : (nf & ~(8 | 32));     // (a pass that went through: the next failure re-learns again)
                    fl = wv::keepi(nf);
                    redo = wv::ballot(retry);
                    ACME_T(TB_GLUE);
                    if (ACME_RARE(redo != 0ull)) break;
                } while ((actm = wv::ballot((fl & 1) != 0)) != 0ull);
                if (ACME_USUAL(redo == 0ull)) break;
                const bool off = retry && (fl & 8) != 0;
                const bool okp = relearn_c(z, retry, wv::ballot(off) != 0ull, off);
                // (exactly singular S: setlhs! fails and the solve returns at once, src/solvers.jl:223-224 -- that pass counts)
                its += (retry && !okp) ? 1 : 0;
                fl = wv::keepi(retry ? (okp ? (fl | 32) : ((fl & ~3) | ((fl & 64) ? 2 : 0))) : fl);
            }
            const bool accepted = (fl & 4) != 0;
            // the accepted iterate's z_L: what the linear rows give for its z_N (the lanes' last "residual" is exactly
            // -(zp_L - W z_N)) -- unless it was accepted off the subspace, as it stood
            const bool offacc = (fl & 16) != 0;
            z = sel(accepted && islin && !offacc, -res, z);
            osub = accepted ? (offacc ? 1 : 0) : osub;
            opos = accepted ? pf[2] : opos;
            lz = sel(accepted, z, lz);
            lp = sel(accepted, target, lp);
        }
        return (fl & 6) != 0;
    };

    // solve(::CachingSolver, p) (src/solvers.jl:347-396) around the base solve, with a bounded
    // store: if one of the (at most CACHE) stored solutions lies strictly nearer to p than the
    // current extrapolation origin, it becomes the origin -- set_extrapolation_origin(base, p_c,
    // z_c) re-linearises there (:183-189) -- and a converged base solve that needed more than 5
    // iterations is stored, overwriting the oldest entry once the store is full (the reference
    // keeps all of them in a k-d tree).  Lane e owns entry e.  Capacity matters for the cells of the
    // bench grid that are hard in steady state: their oracle costs 9.4 Newton iterations per sample
    // without a cache, 4.5 with 8 entries, 3.02 with 16, 3.00 with 32 and 3.02 unbounded -- and a
    // launch lasts as long as its slowest wave.
    const bool caching = A.solver == SOLVER_CACHING_HOMOTOPY;
    auto cached_solve = [&](double target, bool need, int &its) ACME_LAMBDA -> bool {
        double *cp = cch, *cz = czg;
        int *meta = S::META_SCR ? reinterpret_cast<int *>(meta_scr) : reinterpret_cast<int *>(cch + NP * CACHE);   // count, head
        bool reorig = stale != 0;   // (lp, lz) not linearised (in this row order): launch start, ..., or new origin below
        if (caching) {
            const int count = meta[0];
            const double dl = (lig < NP) ? target - lp : 0.0;
            const double best = wv::allsum16(dl * dl);
            double d = 0.0;
            if constexpr (!S::FUSE || S::CHAINWAIT) {
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    const double t = cp[j * CACHE + (lig & (CACHE - 1))] - wv::bcast16<j>(target);
                    d = fma(t, t, d);
                });
            } else {   // cp_j - p_j as a fused broadcast multiply-add with -1 (exact), then the square
                // (all the stored p's requested first: load - wait - use per column exposed the LDS latency
                // np times a sample)
                double cpv[NPr];
                sfor<0, NP>([&](auto jc) ACME_LAMBDA { cpv[decltype(jc)::value] = cp[decltype(jc)::value * CACHE + (lig & (CACHE - 1))]; });
                wv::sched_fence();
                const double m1 = wv::keep(-1.0);
                const double tg = wv::settle(target);      // (two wait states old whatever the compiler did to `target` just before)
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    double t = cpv[j];
                    wv::fmac_bcast<j>(t, tg, m1);
                    d = fma(t, t, d);
                });
            }
            d = (lig < count) ? d : (double)INFINITY;
            const double m = wv::allmin16(d);
            const bool hit = need && count > 0 && m < best;
            if (ACME_RARE(wv::ballot(hit))) {
                const unsigned long long bal = wv::ballot(d == m);
                const int idx = wv::ffs32((int)((bal >> (grp * GROUP)) & 0xFFFFull)) - 1;   // first nearest entry
                const int e = hit ? idx : 0;
                const double cpl = cp[(lig < NP ? lig : 0) * CACHE + e];
                const double czl = (valid && caching) ? cz[e * NN + (lig < NN ? lig : 0)] : 0.0;   // HBM, one line
                lp = hit ? cpl : lp;
                lz = hit ? czl : lz;
                if constexpr (COND) osub = hit ? 0 : osub;      // (a stored solution was an accepted iterate: on the subspace)
            }
            reorig = reorig || hit;
        }
        bool c;
        if constexpr (COND) {
            // Phase 0 (rare): set_extrapolation_origin(solver, lp, lz) for the instances with `reorig`.  Phase 1: the solve
            // at `target`.  Both start alike -- pfull, the condensation for ITS potentiometer positions, pf' -- hence one
            // loop with one copy of that code (the phase is wave-uniform and opaque to the optimiser).
            double cmul[NRr + 1], z0m = 0.0;
            bool mv = false, mvpre = false, back = false;
            // (phase 0's own part as a lambda: the same statements written out inside the loop cost 2.7 % of the headline --
            // 301.1 against 292.8 ms, A/B on one box; the optimiser sees the blocks in another order, and the register
            // allocator then copies the condensation's 24 doubles 150 times less per kernel and spills 160 fewer values)
            auto origin_phase = [&]() ACME_LAMBDA {
                // ---- phase 0: pf, pf' and the condensation are the ORIGIN's ----
                if (wv::ballot(reorig) != 0ull) {
                    bool s0;
                    double d0;
                    const bool roff = reorig && osub != 0;
                    const unsigned long long roffany = wv::ballot(roff);
                    const bool mine0 = lin_c(reorig ? lz : z, false, 0ull, reorig, s0, d0, roffany != 0ull, roff);
                    if (ACME_RARE(wv::ballot(mine0 && reorig) != 0ull)) {      // the lanes' order fails at the origin: re-learn it there
                        (void)relearn_c(reorig ? lz : z, mine0 && reorig, roffany != 0ull, roff);
                        (void)lin_c(reorig ? lz : z, false, 0ull, reorig, s0, d0, roffany != 0ull, roff);
                    }
                    opos = reorig ? pf[2] : opos;
                }
                // (the target's pfull in the lanes' PRESENT order -- an adoption may just have changed it)
                double pfo[3], pft[3];
                sfor<0, 3>([&](auto tc_) ACME_LAMBDA { pfo[decltype(tc_)::value] = pf[decltype(tc_)::value]; });
                set_p(wv::settle(target));
                sfor<0, 3>([&](auto tc_) ACME_LAMBDA { pft[decltype(tc_)::value] = pf[decltype(tc_)::value]; pf[decltype(tc_)::value] = pfo[decltype(tc_)::value]; });
                mv = inst_any(need && islin && !(pft[2] == opos));
                if (wv::ballot(mv) != 0ull) {
                    // The potentiometers moved since the origin was taken: z0 = last_z - last_J \ (last_Jp (p - last_p))
                    // (src/solvers.jl:209-215) on the FULL system, by blocks -- the linear rows' right-hand sides
                    // t_L = Jq_L dpfull through A_LL^-1 (u_L), the others' corrected by J_NL u_L (through pfull, as
                    // fq_L u_L), then the origin's recorded reduced elimination with the linear rows riding along.
                    // Jq at the origin from the FULL q (an `off` evaluation at last_z).
                    evaluate_c(lz, true, mv);
                    const wv::pair_t h_ = wv::ld2(&rd.rc[3 * 2 * GROUP]);
                    const double tvT[3] = {islin ? 1.0 : tv[0], islin ? crw : tv[1], islin ? h_.lo * efl1 : tv[2]};
                    double dp[3], tt = 0.0;
                    sfor<0, 3>([&](auto tc_) ACME_LAMBDA {
                        constexpr int t = decltype(tc_)::value;
                        dp[t] = pft[t] - pf[t];
                        tt = fma(tvT[t], dp[t], tt);
                    });
                    double uL = 0.0;
                    wv::fmac_bcast_chain<NL, true>(uL, tt, cd[1]);
                    const double nu = -uL;
                    double fql[NT][NLr + 1];
                    sfor<0, NT>([&](auto tc_) ACME_LAMBDA {
                        constexpr int t = decltype(tc_)::value;
                        sfor<0, (NL + 1) / 2>([&](auto jc) ACME_LAMBDA {
                            constexpr int j = 2 * decltype(jc)::value;
                            const wv::pair_t v = wv::ld2(&Ms[L.fqr + L.gat(t, j, 0, NN) + 2 * grow]);
                            fql[t][j] = v.lo;
                            fql[t][j + 1] = v.hi;
                        });
                    });
                    double tN = 0.0;
                    sfor<0, 3>([&](auto tc_) ACME_LAMBDA {
                        constexpr int t = decltype(tc_)::value;
                        wv::fmac_bcast_chain<NL, true>(dp[t], nu, fql[t]);
                        tN = fma(tv[t], dp[t], tN);
                    });
                    tt = islin ? uL : tN;
                    double cm[NRr + 1];
                    LU::template load_stored_n<NR, S>(cm, ojp);
                    LU::template apply_loaded_from<NL, NR>(tt, cm);
                    z0m = lz - tt;
                }
            };
            // an origin to (re-)linearise is known before the loop (a stale slab, a solution-cache hit: 13 ... 17 % of an
            // instance's samples on the headline grid): straight into phase 0 -- the target's pfull is formed there anyway;
            // every instance then takes part in the origin's condensation check (theirs is up to date).  One set_p less on
            // those samples: 274.6 -> 272.2 ms on the headline (round 5)
            const bool direct0 = wv::ballot(reorig) != 0ull;
            int ph = wv::opaque(1);
            if (direct0) {
                ph = wv::opaque(0);
                back = true;
            }
            for (;;) {
                // (the target's pfull is formed a second time after phase 0; handing phase 0's copy over instead was measured:
                // six more registers live across the phase -- 283.4 against 272.1 ms)
                set_p(wv::settle(ph != 0 ? target : lp));
                if (ph != 0 && !back) {
                    // an origin to (re-)linearise, or potentiometers that moved since the origin was taken (opos is not
                    // known yet where the origin is stale: settled in phase 0): the origin's turn first
                    mvpre = inst_any(need && islin && !(pf[2] == opos));
                    if (ACME_RARE(wv::ballot(reorig || mvpre) != 0ull)) {
                        ph = wv::opaque(0);
                        back = true;
                        continue;
                    }
                }
                const bool part = ph != 0 ? need : (reorig || mvpre || direct0);
                const bool chg = inst_any(part && islin && !(pf[2] == cpos));
                if (ACME_RARE(wv::ballot(chg) != 0ull)) condense(chg);
                prep_reduced();
                if (ACME_USUAL(ph != 0)) break;
                origin_phase();
                ph = wv::opaque(1);
            }
            // (the origin's recorded multipliers, requested only now: held across the phase loop they cost 16 registers
            // there -- 170 more spilled values in the rare paths -- for latency the other wave covers anyway)
            LU::template load_stored_n<NR, S>(cmul, ojp);
            c = base_solve_c(target, need, its, cmul, mv, z0m);
        } else {
            if (ACME_RARE(wv::ballot(reorig))) {
                set_p(lp);
                bool f0, k0, s0;
                double d0;
                linearize(reorig ? lz : z, false, 0ull, reorig, f0, k0, s0, d0);
            }
            c = base_solve(target, need, its);
        }
        if (caching) {
            const bool keep = need && c && its > 5;
            if (ACME_RARE(wv::ballot(keep))) {
                const int count = meta[0], head = meta[1];
                wv::wave_fence();
                if (keep) {
                    const int slot = count < CACHE ? count : head;
                    if (lig < NP) cp[lig * CACHE + slot] = target;
                    if (valid && lig < NN) cz[slot * NN + lig] = z;   // HBM, entry-major: one coalesced line
                    if (lig == 0) {
                        meta[0] = count < CACHE ? count + 1 : count;
                        meta[1] = count < CACHE ? head : (head + 1) & (CACHE - 1);
                    }
                }
                wv::wave_fence();
            }
        }
        return c;
    };

    // ---- MODE_JAC: get_extrapolation_jacobian for every instance ---------------------------
    // -(J \\ Jp) at the extrapolation origin (last_p, last_z) of sub-problem A.solve_sub
    // (src/solvers.jl:198-201; used by linearize, :407-414): re-linearise there exactly like
    // set_extrapolation_origin does (:183-196) with the columns of Jp riding along in the
    // elimination, in the reference's pivot order if the rows' current order fails the threshold.
    // Nothing of the batch's state is modified.
    if constexpr (MODE == MODE_JAC) {
        sfor<0, NSUB>([&](auto sc) ACME_LAMBDA {
            constexpr int s = decltype(sc)::value;
            if (!(NN > 0 && s < nsub) || s != A.solve_sub) return;
            if (S::NSUB > 1) enter_sub(sc);
            set_p(lp);
            double jp[NPr], dz;
            int phase = 0;
            bool relearn = false, mine = false;
            for (;;) {
                phase = wv::opaque(phase);
                (void)evaluate(lz);
                if (phase == 1) {
                    int orig;
                    (void)LU::pivot_order(a, orig, lig, grp);
                    orig = relearn ? orig : lig;
                    adopt(orig);
                    phase = 2;
                    continue;
                }
                calc_jp(jp);
                dz = res;
                double dinv;
                const unsigned long long viol = LU::template solve_inplace<NP, false, S, false, true>(a, dz, jp, ojp, false, dinv);
                mine = ((viol >> (grp * GROUP)) & 0xFFFFull) != 0ull;
                if (viol != 0ull && phase == 0) {
                    relearn = mine;
                    phase = 1;
                    continue;
                }
                break;
            }
            if (valid && lig < A.nn_io)     // column-major nn x np per instance; NaN if J is singular there
                sfor<0, NP>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    if (j < A.np_io) A.jac_out[(inst * A.np_io + j) * A.nn_io + A.zperm[lig]] = mine ? (double)NAN : -jp[j];
                });
        });
        return;
    }

    // ---- report words -------------------------------------------------------------------
    // kept in LDS (one copy per instance, updated by the instance's lane 0 once per sample)
    // rather than in registers of all 16 lanes: they are never needed inside the solver loop
    long long *rbuf = reinterpret_cast<long long *>(ybuf + S::YBUF);
    double it_total = 0.0;     // Shape::ITREG: this launch's iteration total (exact in a double) and maximum, in registers
    int it_max = 0;
    int dead = valid ? 0 : 1;  // dead: the reference would have thrown at first_nonfinite (an integer, like `stale`)
    if (valid) {
        const long long *rp = A.report + inst * RW_WORDS;
        if (lig < RW_WORDS) rbuf[lig] = rp[lig];
        dead = rp[RW_FIRST_NONFINITE] >= 0 ? 1 : 0;
    }
    wv::wave_fence();

    // ---- time loop ----------------------------------------------------------------------
    constexpr bool solve_mode = MODE == MODE_SOLVE;
    const long long T = solve_mode ? 1 : A.T;
    const int nu_io = A.nu_io, ny_io = A.ny_io;
    // u tile: fetched for chunk 0 before the loop and for chunk c+1 at the end of the LAST sample of
    // chunk c (after its y/x update, the last readers of the old tile).  The ~1 us of HBM latency is
    // exposed once per 16 samples (< 0.5 % of their run time); issuing the loads earlier kept the
    // staging registers alive across the update and cost more in spill traffic than it hid (and on the
    // small shapes, which have the registers, requesting the tile a whole chunk ahead gained nothing).
    // (Global pointers are recomputed here, once per 16 samples, for the same reason.)
    double upre[S::NUR];
    long long u_seen = 0;         // (streamed host run: the last value of *A.u_ready this wave has read)
    auto fetch_u = [&](long long n0) ACME_LAMBDA {
        const double *ug = A.u + ((valid ? inst : 0) * (A.u_stride ? A.u_stride : T) + n0) * nu_io;
        long long cnt = T - n0;
        if (cnt > S::CH) cnt = S::CH;
        if constexpr (MODE == MODE_RUN_STREAM) {
            // streamed host run: the tile may still be on its way into HBM -- wait for the host's word that it has
            // landed (a wave gets ahead of the copy engine only at the very start of a run).  The loads below are
            // ordinary ones: the launcher lays the staging buffer out so that no cache line holds both samples that
            // have landed and samples still to come (rows on 128-byte boundaries, chunks of whole lines), and nothing
            // reads a sample before the word covers it; the acquire orders them after the word all the same.
            if (u_seen < n0 + cnt) {
                while ((u_seen = wv::load_system(A.u_ready)) < n0 + cnt) wv::nap();
                wv::acquire_system();
            }
        }
        sfor<0, NU>([&](auto ic) ACME_LAMBDA {
            constexpr int i = decltype(ic)::value;
            int e = lig + GROUP * i;
            upre[i] = (valid && e < (int)cnt * nu_io) ? ug[e] : 0.0;
        });
    };
    auto stage_u = [&]() ACME_LAMBDA {
        wv::wave_fence();
        sfor<0, NU>([&](auto ic) ACME_LAMBDA {
            constexpr int i = decltype(ic)::value;
            int e = lig + GROUP * i;
            if (e < S::CH * nu_io) ubuf[(e / nu_io) * NU + (e % nu_io)] = upre[i];
        });
        wv::wave_fence();
    };
    if (NU > 0 && !solve_mode) {
        fetch_u(0);
        stage_u();
    }

    // register-resident rows of the linear part (Shape::LINREG / DQREG)
    double wreg[S::LINREG ? S::NLC : 1], dqreg[S::DQREG ? NX + NU : 1];
    if constexpr (S::LINREG)
        sfor<0, S::NLC>([&](auto cc) ACME_LAMBDA { wreg[decltype(cc)::value] = Mg[L.lin(decltype(cc)::value, 0, NX, NU) + lig]; });
    if constexpr (S::DQREG)
        sfor<0, NX + NU>([&](auto cc) ACME_LAMBDA { dqreg[decltype(cc)::value] = Mg[L.sub0 + L.pq(decltype(cc)::value, 0, NP, NX) + lig]; });
    // (arrived HERE: no wait for them inside the time loop, where a u tile may be in flight)
    if constexpr (S::LINREG)
        sfor<0, S::NLC>([&](auto cc) ACME_LAMBDA { wreg[decltype(cc)::value] = wv::keep(wreg[decltype(cc)::value]); });
    if constexpr (S::DQREG)
        sfor<0, NX + NU>([&](auto cc) ACME_LAMBDA { dqreg[decltype(cc)::value] = wv::keep(dqreg[decltype(cc)::value]); });

    for (long long n0 = 0; n0 < T; n0 += S::CH) {
        int cnt = (int)((T - n0 < S::CH) ? (T - n0) : S::CH);
        for (int m = 0; m < cnt; ++m) {
            const long long n = A.sample_base + n0 + m;
            // this sample's inputs, read from the tile ONCE (p and the y/x update both use them)
            double us[NU > 0 ? NU + 1 : 1];
            if constexpr (L.pairs && NU % 2 == 0 && NU > 0) {
                sfor<0, NU / 2>([&](auto kc) ACME_LAMBDA {
                    constexpr int k = 2 * decltype(kc)::value;
                    const wv::pair_t v = wv::ld2(&ubuf[m * NU + k]);
                    us[k] = v.lo;
                    us[k + 1] = v.hi;
                });
            } else {
                sfor<0, NU>([&](auto kc) ACME_LAMBDA { us[decltype(kc)::value] = ubuf[m * NU + decltype(kc)::value]; });
            }
            ACME_T(TB_POST);
            // the nonlinear sub-problems, one after another: later ones see the solutions of the
            // earlier ones through fqprev (src/ACME.jl:675-697)
            sfor<0, NSUB>([&](auto sc) ACME_LAMBDA {
                constexpr int s = decltype(sc)::value;
                if (!(NN > 0 && s < nsub)) return;
                if (solve_mode && s != A.solve_sub) return;
                if (S::NSUB > 1) enter_sub(sc);
                const bool alive = dead == 0;
                // p = dq*x + eq*u + fqprev*z  (src/ACME.jl:678-686)
                double p = 0.0;
                double dqe[NX + NU + 1];       // row lig of [dq | eq], two columns per LDS read where stored in pairs
                if constexpr (L.pairs) {
                    sfor<0, (NX + NU + 1) / 2>([&](auto cc) ACME_LAMBDA {
                        constexpr int c = 2 * decltype(cc)::value;
                        const wv::pair_t v = wv::ld2(&Ms[L.pq(c, 0, NP, NX) + 2 * lig]);
                        dqe[c] = v.lo;
                        dqe[c + 1] = v.hi;
                    });
                } else if constexpr (S::DQREG) {
                    sfor<0, NX + NU>([&](auto cc) ACME_LAMBDA { dqe[decltype(cc)::value] = dqreg[decltype(cc)::value]; });
                } else {
                    sfor<0, NX + NU>([&](auto cc) ACME_LAMBDA { dqe[decltype(cc)::value] = Ms[L.pq(decltype(cc)::value, 0, NP, NX) + lig]; });
                }
                if constexpr (S::FUSE) {       // x: last sample's update
                    sfor<0, NXS>([&](auto sc) ACME_LAMBDA {
                        constexpr int xs = decltype(sc)::value;
                        wv::fmac_bcast_chain<(NX - xs * GROUP < GROUP ? NX - xs * GROUP : GROUP), true, xs * GROUP>(p, x[xs], dqe);
                    });
                } else {
                    sfor<0, NX>([&](auto jc) ACME_LAMBDA {
                        constexpr int j = decltype(jc)::value;
                        double xj = wv::bcast16<j % GROUP>(x[j / GROUP]);
                        p = fma(dqe[j], xj, p);
                    });
                }
                sfor<0, NU>([&](auto kc) ACME_LAMBDA {
                    constexpr int k = decltype(kc)::value;
                    p = fma(dqe[NX + k], us[k], p);
                });
                sfor<0, s>([&](auto pc) ACME_LAMBDA {        // earlier sub-problems' z
                    constexpr int sp = decltype(pc)::value;
                    sfor<0, NN>([&](auto jc) ACME_LAMBDA {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (!S::FUSE) p = fma(Ms[L.fqprev + (sp * NN + j) * NP + lig], wv::bcast16<j>(zs[sp]), p);
                    });
                    if constexpr (S::FUSE) {
                        double fp[NNr];
                        sfor<0, NN>([&](auto jc) ACME_LAMBDA { fp[decltype(jc)::value] = Ms[L.fqprev + (sp * NN + decltype(jc)::value) * NP + lig]; });
                        wv::fmac_bcast_chain<NN, true>(p, zs[sp], fp);
                    }
                });
                if (solve_mode) p = (valid && lig < A.np_io) ? A.p_in[inst * A.np_io + lig] : 0.0;
                // solve(::HomotopySolver, p) (src/solvers.jl:268-296) as a per-instance
                // state machine; every base solve is shared by the wave
                // (need / hasconverged as bits 0 / 1 of an integer in a vector register: see base_solve)
                int hf = wv::keepi(alive ? 1 : 0);
                int mode = 0, its_sample = 0;
                double ha = 0.5, hbest = 0.0, startp = 0.0, target = p;
                ACME_DBG("sample %lld sub %d lane %d p %.17g x %.17g lp %.17g lz %.17g", n, s, lane, p, x[0], lp, lz);
                ACME_T(TB_PRE);
                // The direct attempt first, as straight-line code: it almost always settles the sample, and outside
                // a loop it needs none of the register copies and flag traffic of a loop head.  The bisection loop --
                // a second, cold copy of the solver -- runs only when some instance's direct attempt failed.
                auto hstep = [&](bool need, bool c, int nh) ACME_LAMBDA -> int {     // bookkeeping after one base solve
                    bool direct = need && mode == 0;
                    bool homot = need && mode == 1;
                    bool start = direct && !c;
                    startp = sel(start, lp, startp);
                    bool hgood = homot && c;
                    hbest = sel(hgood, ha, hbest);
                    double new_a = (ha + hbest) / 2.0;
                    bool hbreak = homot && !c && !(hbest < new_a && new_a < ha);
                    ha = sel(hgood, 1.0, sel(homot && !c, new_a, ha));
                    ha = sel(start, 0.5, ha);
                    hbest = sel(start, 0.0, hbest);
                    mode = sel(start, 1, mode);
                    need = need && !(direct && c) && !hbreak && !(homot && hbest >= 1.0);
                    double pa = startp * (1.0 - ha);
                    pa = pa + ha * p;
                    target = sel(need, pa, target);
                    return need ? (nh | 1) : (nh & ~1);
                };
                if constexpr (S::ONELOOP) {
                // ONE copy of the solver: the direct attempt is the first pass of the homotopy loop
                do {
                    const bool need = (hf & 1) != 0;
                    int its;
                    const bool c = cached_solve(target, need, its);
                    its_sample += need ? its : 0;
                    int nh = need ? ((hf & ~2) | (c ? 2 : 0)) : hf;
                    if (ACME_USUAL(A.solver == SOLVER_SIMPLE || !wv::ballot(need && !(mode == 0 && c)))) nh &= ~1;
                    else nh = hstep(need, c, nh);
                    hf = wv::keepi(nh);
                    ACME_T(TB_HOMO);
                } while (ACME_RARE(wv::ballot((hf & 1) != 0)));
                } else {
                {
                    const bool need = alive;
                    int its;
                    const bool c = cached_solve(target, need, its);
                    its_sample = need ? its : 0;
                    int nh = need ? ((hf & ~2) | (c ? 2 : 0)) : hf;
                    if (ACME_USUAL(A.solver == SOLVER_SIMPLE || !wv::ballot(need && !c))) {
                        nh &= ~1;
                    } else {
                        nh = hstep(need, c, nh);
                    }
                    hf = wv::keepi(nh);
                    ACME_T(TB_HOMO);
                }
                while (ACME_RARE(wv::ballot((hf & 1) != 0))) {
                    bool need = (hf & 1) != 0;
                    int its;
                    bool c = cached_solve(target, need, its);
                    its_sample += need ? its : 0;
                    int nh = need ? ((hf & ~2) | (c ? 2 : 0)) : hf;
                    if (A.solver == SOLVER_SIMPLE || !wv::ballot(need && !(mode == 0 && c))) nh &= ~1;
                    else nh = hstep(need, c, nh);
                    hf = wv::keepi(nh);
                    ACME_T(TB_HOMO);
                }
                }
                const bool conv = (hf & 2) != 0;
                zs[s] = alive ? z : 0.0;
                if (solve_mode) {   // hand the solver's answer back; no y, no state update
                    if (valid && lig < A.nn_io) A.z_out[inst * A.nn_io + A.zperm[lig]] = z;
                    if (valid && lig == 0) {
                        A.conv_out[inst] = conv ? 1 : 0;
                        A.iters_out[inst] = its_sample;
                    }
                } else {
                    // convergence policy of step! (src/ACME.jl:688-694)
                    bool failed = alive && !conv;
                    if (ACME_RARE(wv::ballot(failed))) {
                        unsigned long long nf = S::LITROWS ? wv::ballot(!(z * 0.0 == 0.0)) & rows4((1ull << NN) - 1ull)
                                                           : wv::ballot(lig < NN && !(z * 0.0 == 0.0));
                        bool zfinite = ((nf >> (grp * GROUP)) & 0xFFFFull) == 0ull;
                        bool warn = failed && zfinite;
                        bool die = failed && !zfinite;
                        if (lig == 0 && warn) {
                            rbuf[RW_NWARN] += 1;
                            if (rbuf[RW_FIRST_NONCONV] < 0) rbuf[RW_FIRST_NONCONV] = n;
                        }
                        if (lig == 0 && die && rbuf[RW_FIRST_NONFINITE] < 0) rbuf[RW_FIRST_NONFINITE] = n;
                        dead = die ? 1 : dead;
                    }
                    if constexpr (S::ITREG) {      // (every lane of the instance keeps the same two numbers: no EXEC games)
                        const int itn = alive ? its_sample : 0;
                        it_total += (double)itn;
                        it_max = itn > it_max ? itn : it_max;
                    } else if (S::LITROWS ? and_rows<1ull, true>(alive) : (lig == 0 && alive)) {   // fire-and-forget LDS atomics: no round trip to wait for
                        wv::lds_add(&rbuf[RW_ITERS_TOTAL], (long long)its_sample);
                        wv::lds_max(&rbuf[RW_ITERS_MAX], (long long)its_sample);
                    }
                }
                if (S::NSUB > 1) leave_sub(sc);
            });
            if (solve_mode) continue;
            const bool refill = NU > 0 && m == cnt - 1 && n0 + S::CH < T;
            const bool live = dead == 0;
            wv::sched_fence();
            constexpr int LD = NX + NY;
            if constexpr (NX + NY <= GROUP && NX > 0) {
                // y = y0 + dy*x + ey*u + fy*z with the OLD x (src/ACME.jl:699-706) and
                // x = x0 + a*x + b*u + c*z (:708-714) in ONE pass: lanes < NX hold the rows of
                // [a b c x0], lanes NX .. NX+NY-1 the rows of [dy ey fy y0] (same operations per row
                // as two separate passes, half the instructions and LDS reads)
                // the row of [x0 a b c] / [y0 dy ey fy] this lane owns: two columns per LDS read where the
                // layout stores them in pairs (Layout::lin)
                constexpr int NLC = S::NLC;
                double w[NLC + 1];
                if constexpr (L.linp) {
                    sfor<0, (NLC + 1) / 2>([&](auto cc) ACME_LAMBDA {
                        constexpr int c = 2 * decltype(cc)::value;
                        const wv::pair_t v = wv::ld2(&M[L.lin(c, 0, NX, NU) + 2 * lig]);
                        w[c] = v.lo;
                        w[c + 1] = v.hi;
                    });
                } else if constexpr (S::LINREG) {
                    sfor<0, NLC>([&](auto cc) ACME_LAMBDA { w[decltype(cc)::value] = wreg[decltype(cc)::value]; });
                } else {
                    sfor<0, NLC>([&](auto cc) ACME_LAMBDA { w[decltype(cc)::value] = M[L.lin(decltype(cc)::value, 0, NX, NU) + lig]; });
                }
                double acc = w[0];
                if constexpr (S::FUSE) {
                    wv::fmac_bcast_chain<NX, S::CHAINWAIT, 1>(acc, x[0], w);
                } else {
                    sfor<0, NX>([&](auto jc) ACME_LAMBDA {
                        constexpr int j = decltype(jc)::value;
                        acc = fma(w[1 + j], wv::bcast16<j>(x[0]), acc);
                    });
                }
                sfor<0, NU>([&](auto kc) ACME_LAMBDA {
                    constexpr int k = decltype(kc)::value;
                    acc = fma(w[1 + NX + k], us[k], acc);
                });
                sfor<0, S::NSUB>([&](auto sc) ACME_LAMBDA {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (S::FUSE) {       // (zs[] was selected just above: the statement waits)
                        wv::fmac_bcast_chain<NN, true, 1 + NX + NU + s * NN>(acc, zs[s], w);
                    } else {
                        sfor<0, NN>([&](auto jc) ACME_LAMBDA {
                            constexpr int j = decltype(jc)::value;
                            acc = fma(w[1 + NX + NU + s * NN + j], wv::bcast16<j>(zs[s]), acc);
                        });
                    }
                });
                if (NY > 0 && lig >= NX && lig < NX + NY) ybuf[m * NY + lig - NX] = live ? acc : (double)NAN;
                x[0] = sel(S::LITROWS ? and_rows<(1ull << NX) - 1ull, true>(live) : (live && lig < NX), acc, x[0]);
            } else {
            // y = y0 + dy*x + ey*u + fy*z  with the OLD x  (src/ACME.jl:699-706)
            if (NY > 0) {
                double yy = M[L.y0 + lig];
                sfor<0, NX>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    yy = fma(M[L.dy + j * LD + lig], wv::bcast16<j % GROUP>(x[j / GROUP]), yy);
                });
                sfor<0, NU>([&](auto kc) ACME_LAMBDA {
                    constexpr int k = decltype(kc)::value;
                    yy = fma(M[L.ey + k * LD + lig], us[k], yy);
                });
                sfor<0, S::NSUB>([&](auto sc) ACME_LAMBDA {
                    constexpr int s = decltype(sc)::value;
                    sfor<0, NN>([&](auto jc) ACME_LAMBDA {
                        constexpr int j = decltype(jc)::value;
                        yy = fma(M[L.fy + (s * NN + j) * LD + lig], wv::bcast16<j>(zs[s]), yy);
                    });
                });
                if (lig < NY) ybuf[m * NY + lig] = live ? yy : (double)NAN;
            }
            wv::sched_fence();
            // x = x0 + a*x + b*u + c*z  (src/ACME.jl:708-714)
            if (NX > 0) {
                double xn[NXSr];
                sfor<0, NXS>([&](auto sc) ACME_LAMBDA {
                    constexpr int s = decltype(sc)::value;
                    xn[s] = M[L.x0 + s * GROUP + lig];
                });
                sfor<0, NX>([&](auto jc) ACME_LAMBDA {
                    constexpr int j = decltype(jc)::value;
                    double xj = wv::bcast16<j % GROUP>(x[j / GROUP]);
                    sfor<0, NXS>([&](auto sc) ACME_LAMBDA {
                        constexpr int s = decltype(sc)::value;
                        xn[s] = fma(M[L.a + j * LD + s * GROUP + lig], xj, xn[s]);
                    });
                });
                sfor<0, NU>([&](auto kc) ACME_LAMBDA {
                    constexpr int k = decltype(kc)::value;
                    double uk = us[k];
                    sfor<0, NXS>([&](auto sc) ACME_LAMBDA {
                        constexpr int s = decltype(sc)::value;
                        xn[s] = fma(M[L.b + k * LD + s * GROUP + lig], uk, xn[s]);
                    });
                });
                sfor<0, S::NSUB>([&](auto pc) ACME_LAMBDA {
                    constexpr int sp = decltype(pc)::value;
                    sfor<0, NN>([&](auto jc) ACME_LAMBDA {
                        constexpr int j = decltype(jc)::value;
                        double zj = wv::bcast16<j>(zs[sp]);
                        sfor<0, NXS>([&](auto sc) ACME_LAMBDA {
                            constexpr int s = decltype(sc)::value;
                            xn[s] = fma(M[L.c + (sp * NN + j) * LD + s * GROUP + lig], zj, xn[s]);
                        });
                    });
                });
                sfor<0, NXS>([&](auto sc) ACME_LAMBDA {
                    constexpr int s = decltype(sc)::value;
                    x[s] = sel(live, xn[s], x[s]);
                });
            }
            }
            if (ACME_RARE(refill)) {     // next u tile: this sample's y/x update was the last reader of the old one
                fetch_u(n0 + S::CH);
                stage_u();
            }
        }
        // flush the y tile, coalesced
        if (NY > 0 && !solve_mode) {
            double *yg = A.y + ((valid ? inst : 0) * (A.y_stride ? A.y_stride : T) + n0) * ny_io;
            wv::wave_fence();
            for (int e = lig; e < cnt * ny_io; e += GROUP)
                if (valid) yg[e] = ybuf[(e / ny_io) * NY + (e % ny_io)];
            wv::wave_fence();
        }
    }

#ifdef ACME_TIMING
    ACME_T(TB_POST);
    if (valid && lig == 0 && !solve_mode && T >= TB_N && NY > 0)   // bucket totals replace the first samples of y
        for (int i = 0; i < TB_N; ++i) A.y[inst * T * ny_io + (long long)i * ny_io] = (double)tb[i];
#endif
    // ---- write back state and report ----------------------------------------------------
    wv::wave_fence();
    if (valid) {
        double *st = A.state + inst * S::STATE;
        sfor<0, NXS>([&](auto sc) ACME_LAMBDA {
            constexpr int s = decltype(sc)::value;
            int i = s * GROUP + lig;
            if (i < NX) st[i] = x[s];
        });
        if (S::NSUB == 1) leave_sub(std::integral_constant<int, 0>{});
        sfor<0, NSUB>([&](auto sc) ACME_LAMBDA {
            constexpr int s = decltype(sc)::value;
            if (NP > 0 && lig < NP) st[NX + s * NP + lig] = lps[s];
            if (NN > 0 && lig < NN) st[NX + NSUB * NP + s * NN + lig] = lzs[s];
            A.roworder[(inst * NSUB + s) * GROUP + lig] = rowids[s];
        });
        if (A.solver == SOLVER_CACHING_HOMOTOPY)
            for (int s = 0; s < S::NSUBr; ++s)
                for (int i = lig; i < S::CACHEPM; i += GROUP)
                    A.cache[inst * S::CACHEIH + s * S::CACHE1H + i] =
                        (i < NP * CACHE || !S::META_SCR) ? cache0[s * S::CACHE1 + i] : (i == NP * CACHE ? meta_scr[0] : 0.0);
        if constexpr (S::ITREG) {
            if (lig == 0 && !solve_mode) {
                rbuf[RW_ITERS_TOTAL] += (long long)it_total;
                rbuf[RW_ITERS_MAX] = rbuf[RW_ITERS_MAX] > it_max ? rbuf[RW_ITERS_MAX] : (long long)it_max;
            }
            wv::wave_fence();
        }
        if (lig < RW_WORDS && !solve_mode) A.report[inst * RW_WORDS + lig] = rbuf[lig];
        if constexpr (COND) if (lig == 0) A.cflags[inst] = osub;
    }
}
#undef grow

}  // namespace acme
