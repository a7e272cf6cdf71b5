// acme_coop.h -- the mid-size kernel: models beyond the tuned shapes (17 ... 64 unknowns in ONE nonlinear sub-problem, up to
// 64 parameters / 128 q rows / any number of states that fits) run COOPERATIVELY -- one circuit instance per DPP row of 16
// lanes, the rows of every matrix and vector dealt out over the lanes (row r to lane r mod 16), every working array of the
// instance in LDS -- instead of one lane per instance with its working arrays in HBM (acme_generic.h: 2.7e6
// instance*samples/s at 20 unknowns, about the CPU's rate).  The reference's LU "for sizes up to about 60 x 60"
// (src/solvers.jl:53-54) is exactly this range, and nldecompose! hands such sub-problems over whenever a circuit's
// nonlinearity does not decompose (src/ACME.jl:349-378).
//
// Same data (GenHeader / GArgs / row tables / state / caches), same solver stack, same ARITHMETIC per matrix entry as the
// generic kernel -- which restates the reference operation for operation:
//   step!           src/ACME.jl:666-715      closures       src/ACME.jl:176-194,236-252
//   LinearSolver    src/solvers.jl:46-132    SimpleSolver   src/solvers.jl:151-236
//   HomotopySolver  src/solvers.jl:247-302   CachingSolver  src/solvers.jl:319-396 (bounded store, as everywhere here)
// so outputs and iteration counts are the generic kernel's (and the oracle's) -- only WHO computes an entry differs.
// Two instantiations (GArgs::coop_nc):
//   * any size (33 ... 64 unknowns; ACME_COOP_LITERAL=1: every size): every working array of an instance in LDS at the generic
//     kernel's offsets; the elimination walks through LDS, row interchanges are real and kept as the composed gather src[]
//     (x_permuted[i] = x[src[i]]), which is what solve! applies first (src/solvers.jl:103-109) -- the reference's arithmetic
//     entry by entry, outputs bit-identical to the lane-per-instance kernel's;
//   * 17 ... 32 unknowns (one kernel per column count, rounded up to four): the Jacobian's rows live in REGISTERS from
//     evaluate! to the Newton step, two per lane, and are eliminated the way the tuned 16-lane kernels do it -- in the row
//     order the instance has learnt, without a pivot search while every multiplier stays below a threshold, Gauss-Jordan
//     with fused DPP multiply-adds, the Newton step riding along: see below ("the THRESHOLD path").  The reference's
//     pivoting (coop_lu_rows) runs only to LEARN a new order.  Results agree with the reference's to rounding (the stated
//     deviation a-13 of DESIGN.md), iteration totals are the oracle's on every parity case.
// The waves of a block share one staged copy of the row tables and, if the batch shares its model image, of the image
// (GArgs::coop_wpb / coop_gpw / coop_imgl: the launch shape, chosen per model by the host, acme_api.inc coop_shape).
//
// Control flow is WAVE-UNIFORM throughout (the four instances of a wave iterate together, finished ones ride along with
// their writes predicated off), as in the tuned kernels: data-dependent trip counts are ballots.
//
// A wave never talks to another wave once the block's shared tables are staged (one barrier).
#pragma once
#include "acme_generic.h"

namespace acme {

constexpr int COOP_MAX_N = 64;      // unknowns / parameters of the sub-problem (4 rows per lane)
constexpr int COOP_SLOTS = COOP_MAX_N / GROUP;
// Instantiation codes (GArgs::coop_nc): > 0 columns of the Jacobian in registers; 0 the reference's LU literally; -1 ... -4 the
// threshold path on a matrix in LDS with that many rows per lane, one instance per 16 lanes; COOP_WAVE64: the same with one
// instance per WAVE, one row per lane -- a wave's time per sample is its instruction count whatever its lanes do (one wave
// issues one instruction per four or five cycles), the instances resident per compute unit are what the LDS holds whatever
// waves carry them, so beyond 32 unknowns (three or four rows per lane: every row operation three or four times in the
// instruction stream) the row-per-lane layout is three to four times faster.
constexpr int COOP_WAVE64 = -64;
ACME_HD constexpr int coop_lpi(int nc) { return nc == COOP_WAVE64 ? 64 : GROUP; }
ACME_HD constexpr int coop_ns(int nc) { return nc == COOP_WAVE64 ? 1 : nc < 0 ? -nc : 0; }

#ifdef ACME_DEV
// -DACME_COOP_TIMING (tools/coop_timing_probe.py): shader-clock cycles per code region, per wave, written over y's first samples
#ifdef ACME_COOP_TIMING
enum { CT_SETP, CT_EXTRAP, CT_EVAL, CT_LU, CT_SOLVE, CT_ACCEPT, CT_LOOKUP, CT_XY, CT_PRE, CT_REST, CT_LU_SEARCH, CT_LU_HAND,
       CT_S_SCAN, CT_S_HEAD, CT_S_BAND, CT_S_REST, CT_S_CHUNKS, CT_S_STEPS, CT_N };
struct CoopTimer { long long t[CT_N]; long long mark; };
#define COOP_T(c, b) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = (long long)__builtin_readcyclecounter(); \
                          (c).tm->t[b] += t_ - (c).tm->mark; (c).tm->mark = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define COOP_T(c, b) do { } while (0)
#endif
#endif  // ACME_DEV
// Where an instance's working arrays sit in its LDS workspace (doubles).  The any-size kernel (nc = 0) uses the generic
// kernel's layout (GenHeader::w_*: every array of the solver stack, two factor matrices with a pitch of nnmax + 3 ... 6).
// The kernels with the running factorisation in REGISTERS (nc = 20 ... 32 columns) keep only what crosses lanes or
// samples: ONE factor matrix (the extrapolation origin's, pitch nc + 2), no Jacobian, residual, Newton step or Jq
// non-zeros (registers), and two nc-double hand-off buffers (the pivot row of an elimination step, the x_j of a sweep) --
// 8.8 KB per instance at 20 unknowns instead of 15.9: 16 resident instances per compute unit instead of 8, a wave for
// every SIMD.
struct CoopOff {
    int x, xn, z, lp, lz, ljp, llu, lsrc, p, pa, sp, zz, res, dz, lu, src, q, pf, tv, tmp, u, prow, xb, ld, dinv, total;
};
ACME_HD inline CoopOff coop_offsets(const GenHeader &H, int nc) {
    CoopOff o{};
    const bool has_sub = H.nsub > 0;
    if (nc == 0) {
        o.x = H.w_x; o.xn = H.w_xn; o.z = H.w_z;
        o.lp = has_sub ? H.sub[0].w_lp : 0; o.lz = has_sub ? H.sub[0].w_lz : 0; o.ljp = has_sub ? H.sub[0].w_ljp : 0;
        o.llu = has_sub ? H.sub[0].w_llu : 0; o.lsrc = has_sub ? H.sub[0].w_lpiv : 0;
        o.p = H.w_p; o.pa = H.w_pa; o.sp = H.w_sp; o.zz = H.w_zz; o.res = H.w_res; o.dz = H.w_dz; o.lu = H.w_lu; o.src = H.w_piv;
        o.q = H.w_q; o.pf = H.w_pf; o.tv = H.w_tv; o.tmp = H.w_tmp; o.u = H.w_u; o.ld = H.ldf; o.total = H.ws_total;
        return o;
    }
    const int nn = H.sub[0].nn, np = H.sub[0].np;
    int w = 0;
    if (nc < 0) {
        // the threshold path on ONE matrix in LDS (coop_lu_lds): row p = position p's Jacobian row, then its factors;
        // column nn the right-hand side; the rows' 1 / pivot apart (a pair of columns is the unit of the elimination's
        // reads and writes: the right-hand side's partner is slack).  Every array at an even offset (16-byte accesses).
        auto take = [&](int n) { const int r = w; w += (n + 1) & ~1; return r; };
        o.ld = ((nn + 3) & ~3) + 2;                  // (>= nn + 2, and 2 mod 4 doubles: the 16 rows of a DPP row in 16 bank groups)
        o.llu = take((nn + 1) * o.ld);               // (one row more: where the lanes beyond the matrix read and write when a wave holds ONE instance)
        o.xb = take(nn);                             // (behind the matrix: the replay's batched reads may run a few doubles past its last row)
        o.dinv = take(nn);
        o.tv = take(4 * nn);
        // Jp by residual ROW, whatever position holds it: np columns and one that stays zero (where the padding entries of the
        // rows' sparse forms point, GenSub::o_pcol) -- or just the kp entries per row of the sparse form (GenHeader::jp_sparse)
        o.ljp = take(H.ell && H.jp_sparse ? nn * H.sub[0].kp : nn * (np + 1));
        o.lsrc = take(nn);
        o.lp = take(np + 1);                         // (... and the p vectors' entry np stays zero likewise)
        o.lz = take(nn);
        o.x = take(H.nx);
        o.xn = take(H.nx);
        o.z = take(H.nnt);
        o.p = take(H.npmax + 1);
        o.pa = take(H.npmax + 1);
        o.sp = take(H.npmax + 1);
        o.zz = take(H.nnmax);
        o.q = take(H.nqmax);
        o.pf = take(H.nqmax);
        o.tmp = take(H.nnmax);
        o.u = take(H.nu);
        o.res = o.dz = o.lu = o.src = o.prow = 0;
        o.total = w;
        return o;
    }
    o.ld = nc + 2;                                   // (2 mod 4 doubles: the 16 rows of a DPP row in 16 bank groups)
    o.llu = w; w += nn * o.ld;                       // 16-byte aligned rows: offset and pitch even
    o.prow = w; w += nc + 2;                         // (the pivot row and, behind it, the position it came from)
    o.xb = w; w += nc;
    o.ljp = w; w += nn * np;
    o.lsrc = w; w += nn;
    o.lp = w; w += np;
    o.lz = w; w += nn;
    o.x = w; w += H.nx;
    o.xn = w; w += H.nx;
    o.z = w; w += H.nnt;
    o.p = w; w += H.npmax;
    o.pa = w; w += H.npmax;
    o.sp = w; w += H.npmax;
    o.zz = w; w += H.nnmax;
    o.q = w; w += H.nqmax;
    o.pf = w; w += H.nqmax;
    o.tmp = w; w += H.nnmax;
    o.u = w; w += H.nu;
    o.res = o.dz = o.lu = o.src = o.tv = 0;          // (registers)
    o.total = w;
    return o;
}
#ifdef ACME_DEV
struct CoopCtx {
    const GArgs &A;
    const GenHeader &H;
    const CoopOff &O;
    const double *M;         // this instance's image: the block's copy in LDS (a shared image) or HBM (private images)
    double *W;               // this instance's workspace in LDS: the generic kernel's offsets (GenHeader::w_*)
    double *Cp;              // ... and its solution cache's stored p's and counters: cp[np][CACHE] | count, head (LDS)
    const double *tk;        // the block's copy of the row tables in LDS: k[8] of every row ([blk][8][16]) ...
    const int *ti;           // ... and the rows' ints ([blk][ROWI][16])
    int lig, grp;
    int lpi;                 // lanes per instance: 16 (one instance per DPP row) or 64 (COOP_WAVE64: one instance per wave)
    bool ell;                // this instantiation reads the sparse forms of the model's matrices (coop_reads_sparse)
    long long i;
    bool valid;
    bool wr;                 // this row of 16 lanes writes to global memory (false: it mirrors the wave's first row, coop_main)
#ifdef ACME_COOP_TIMING
    CoopTimer *tm;
#endif
};
// LDS of one wave (doubles): [ image (shared images only) | row tables | gpw x (workspace | cache p's) ]
ACME_HD inline int coop_table_doubles(const GenHeader &H) {
    const int blocks = (H.nnt + GROUP - 1) / GROUP;
    return blocks * (8 * GROUP + ROWI * GROUP / 2);
}
ACME_HD inline int coop_cache_doubles(const GenHeader &H) { return H.nsub > 0 ? ((H.sub[0].np * CACHE + 2 + 1) & ~1) : 0; }
ACME_HD inline int coop_inst_doubles(const GenHeader &H, int nc) { return ((coop_offsets(H, nc).total + 1) & ~1) + coop_cache_doubles(H); }
// what a block stages of a shared model image: all of it -- or, for the instantiations on a matrix in LDS when the image has
// its sparse forms (GenHeader::ell), only those: they read nothing else
// Who reads what of the image.  The instantiations on a matrix in LDS: the sparse forms only (and stage only those).  The
// register instantiations: the matrices' sparse forms for the matrix-vector products (q, p, x / y: two or three multiply-adds
// instead of a dense row's twenty) and the DENSE fq / pexp for the rows of J and Jp, which they assemble in registers with
// compile-time column numbers -- they stage everything but the rows' tables at the image's end (GenHeader::reg_sparse: where
// that costs no resident instance -- 20 unknowns + 12 %, 32 + 4 %; at 24 it would cost one).  The literal one: the dense
// matrices, as the lane-per-instance kernel it is bit-identical to.
ACME_HD inline bool coop_reads_sparse(const GenHeader &H, int nc) { return (nc < 0 || (nc > 0 && H.reg_sparse != 0)) && H.ell != 0; }
ACME_HD inline int coop_image_first(const GenHeader &H, int nc) { return nc < 0 && H.ell != 0 ? H.o_ell : 0; }
ACME_HD inline int coop_image_doubles(const GenHeader &H, int nc) {
    const int end = nc < 0 && H.ell != 0 ? H.image_total : coop_reads_sparse(H, nc) ? H.o_jtab : H.o_ell;
    return (end - coop_image_first(H, nc) + 1) & ~1;
}
ACME_HD inline int coop_shared_doubles(const GenHeader &H, bool shared_image, int nc) {
    return (shared_image ? coop_image_doubles(H, nc) : 0) + coop_table_doubles(H);
}
ACME_DEV bool coop_any(const CoopCtx &c, bool x) {
    const unsigned long long b = wv::ballot(x);
    return c.lpi == 64 ? b != 0ull : ((b >> (c.grp * GROUP)) & 0xFFFFull) != 0ull;
}
// maximum / minimum over the lanes of an instance, in every one of them
ACME_DEV double coop_allmax(const CoopCtx &c, double v) { return c.lpi == 64 ? wv::allmax64(v) : wv::allmax16(v); }
ACME_DEV double coop_allmin(const CoopCtx &c, double v) { return c.lpi == 64 ? wv::allmin64(v) : wv::allmin16(v); }

// Inner loops in BATCHES: a wave has nothing but its own instruction stream to hide a load's latency behind (LDS ~100
// cycles, the model image in L2 several hundred), and the compiler may not move a load across the store of the iteration
// before it (same array, run-time indices).  So every inner loop first requests COOP_B operands of each kind, then does
// its arithmetic in the reference's order, then stores: one latency per batch instead of one per multiply-add.
constexpr int COOP_B = 8;
// acc + sum_{j < n} a[j * sa] * b[j], accumulated in the order j = 0, 1, ... (fma chain, as the generic kernel's loops)
ACME_DEV double coop_dot(const double *a, int sa, const double *b, int n, double acc) {
    int j = 0;
    for (; j + COOP_B <= n; j += COOP_B) {                 // whole batches: no index clamp, no predicate
        double av[COOP_B], bv[COOP_B];
        for (int u = 0; u < COOP_B; ++u) {
            av[u] = a[(j + u) * sa];
            bv[u] = b[j + u];
        }
        for (int u = 0; u < COOP_B; ++u) acc = fma(av[u], bv[u], acc);
    }
    if (j < n) {                                           // the rest
        double av[COOP_B], bv[COOP_B];
        for (int u = 0; u < COOP_B; ++u) {
            const int jj = j + u < n ? j + u : n - 1;      // (re-reads the last operand: never out of range)
            av[u] = a[jj * sa];
            bv[u] = b[jj];
        }
        for (int u = 0; u < COOP_B; ++u)
            if (j + u < n) acc = fma(av[u], bv[u], acc);
    }
    return acc;
}

// acc + sum_{j < n} a[j * sa] * (b[j] - b0[j]), likewise
ACME_DEV double coop_dot_diff(const double *a, int sa, const double *b, const double *b0, int n, double acc) {
    int j = 0;
    for (; j + COOP_B <= n; j += COOP_B) {
        double av[COOP_B], bv[COOP_B], lv[COOP_B];
        for (int u = 0; u < COOP_B; ++u) {
            av[u] = a[(j + u) * sa];
            bv[u] = b[j + u];
            lv[u] = b0[j + u];
        }
        for (int u = 0; u < COOP_B; ++u) acc = fma(av[u], bv[u] - lv[u], acc);
    }
    if (j < n) {
        double av[COOP_B], bv[COOP_B], lv[COOP_B];
        for (int u = 0; u < COOP_B; ++u) {
            const int jj = j + u < n ? j + u : n - 1;
            av[u] = a[jj * sa];
            bv[u] = b[jj];
            lv[u] = b0[jj];
        }
        for (int u = 0; u < COOP_B; ++u)
            if (j + u < n) acc = fma(av[u], bv[u] - lv[u], acc);
    }
    return acc;
}

// acc + (row r of a matrix in its sparse form, GenEll) * x: the non-zeros in ascending column order -- the dense loop's
// multiply-adds without the ones that add 0 * x_j
ACME_DEV double coop_ell_dot(const double *M, const GenEll &E, int r, const double *x, double acc) {
#ifndef ACME_ELL_BATCH
#define ACME_ELL_BATCH 2          // (entries requested together: a circuit's rows hold one to three)
#endif
    constexpr int B = ACME_ELL_BATCH;
    for (int e = 0; e < E.k; e += B) {
        double v[B], xv[B];
        int ci[B];
        for (int u = 0; u < B; ++u) {
            const int ee = e + u < E.k ? e + u : E.k - 1;
            v[u] = M[E.o_val + ee * E.rows + r];
            ci[u] = (int)M[E.o_col + ee * E.rows + r];
        }
        for (int u = 0; u < B; ++u) xv[u] = x[ci[u]];
        for (int u = 0; u < B; ++u)
            if (e + u < E.k) acc = fma(v[u], xv[u], acc);
    }
    return acc;
}

// a row's descriptor out of the block's LDS copy of the tables (the constants beyond k[0..7], which only the rare element
// kinds read, stay in HBM behind rd.rc)
ACME_DEV void coop_rowdesc(const CoopCtx &c, int R, RowDesc &rd, int (&tc)[4]) {
    const int blk = R / GROUP, ln = R % GROUP;
    const int *ri = c.ti + blk * ROWI * GROUP + ln;
    rd.kind = ri[0 * GROUP];
    rd.erow = ri[1 * GROUP];
    rd.flags = ri[2 * GROUP];
    for (int t = 0; t < 4; ++t) tc[t] = ri[(3 + t) * GROUP];
    rd.rc = c.A.rowc + (long long)blk * ROWC * GROUP + ln;
    for (int k = 0; k < 8; ++k) rd.k[k] = c.tk[(blk * 8 + k) * GROUP + ln];
}

// pfull <- q0 + pexp p  (set_p closure, src/ACME.jl:237-243); p at w_p must be visible (fenced)
ACME_DEV void coop_set_p(const CoopCtx &c, const GenSub &s, int w_p) {
    for (int r = c.lig; r < s.nq; r += c.lpi) {
        c.W[c.O.pf + r] = c.ell ? coop_ell_dot(c.M, s.e_pexp, r, c.W + w_p, c.M[s.o_q0s + r])
                                  : coop_dot(c.M + s.o_pexp + r, s.nq, c.W + w_p, s.np, c.M[s.o_q0 + r]);
    }
    wv::wave_fence();
}

// evaluate!(nleq, z) (src/ACME.jl:178-188, src/circuit.jl:10-17): res, J (ROW-major at o_lu, row pitch GenHeader::ldf), the rows' Jq
// non-zeros.
// Returns (per lane) whether one of its residuals / Jacobian entries is not finite.
ACME_DEV bool coop_evaluate(const CoopCtx &c, const GenSub &s, int w_z, int o_lu) {
    const GenHeader &H = c.H;
    for (int r = c.lig; r < s.nq; r += c.lpi) {
        c.W[c.O.q + r] = c.ell ? coop_ell_dot(c.M, s.e_fq, r, c.W + w_z, c.W[c.O.pf + r])
                                 : coop_dot(c.M + s.o_fq + r, s.nq, c.W + w_z, s.nn, c.W[c.O.pf + r]);
    }
    wv::wave_fence();
    bool bad = false;
    const wv::ExpTab etab = wv::load_exp_tab();          // (once per evaluate!, not twice per row)
    for (int r = c.lig; r < s.nn; r += c.lpi) {
        RowDesc rd;
        int tc[4];
        coop_rowdesc(c, s.row0 + r, rd, tc);
        double e[4], tv[4], res;
        for (int t = 0; t < 4; ++t) e[t] = c.W[c.O.q + tc[t]];
        const bool expo = rd.kind == RK_DIODE || rd.kind == RK_BJT;
        const double exA = exp_junction(expo ? e[0] * rd.k[0] : 0.0, etab);
        const double exB = H.has_bjt ? exp_junction(rd.kind == RK_BJT ? e[1] * rd.k[1] : 0.0, etab) : 1.0;
        eval_row<true, 4>(rd, e, exA, exB, res, tv);
        c.W[c.O.res + r] = res;
        bad = bad || !(res * 0.0 == 0.0);
        for (int t = 0; t < 4; ++t) c.W[c.O.tv + 4 * r + t] = tv[t];
        for (int j = 0; j < s.nn; j += 4) {         // J row = Jq row * fq, four columns' operands at a time
            double fv[4][4];
            for (int u = 0; u < 4; ++u) {
                const int jj = j + u < s.nn ? j + u : s.nn - 1;
                for (int t = 0; t < 4; ++t) fv[u][t] = c.M[s.o_fq + jj * s.nq + tc[t]];
            }
            for (int u = 0; u < 4; ++u) {
                // (a row's Jq non-zeros may name the same q row twice -- padding terms carry a zero derivative)
                double acc = 0.0;
                for (int t = 0; t < 4; ++t) acc = fma(tv[t], fv[u][t], acc);
                if (j + u < s.nn) c.W[o_lu + r * H.ldf + j + u] = acc;
                bad = bad || !(acc * 0.0 == 0.0);
            }
        }
    }
    wv::wave_fence();
    return bad;
}

// calc_Jp closure (src/ACME.jl:246-251) with the Jq of the latest evaluate!, written to w_dst where `pred`
ACME_DEV void coop_calc_jp(const CoopCtx &c, const GenSub &s, int w_dst, bool pred) {
    const GenHeader &H = c.H;
    for (int r = c.lig; r < s.nn; r += c.lpi) {
        RowDesc rd;
        int tc[4];
        coop_rowdesc(c, s.row0 + r, rd, tc);
        double tv[4];
        for (int t = 0; t < 4; ++t) tv[t] = c.W[c.O.tv + 4 * r + t];
        for (int j = 0; j < s.np; j += 4) {
            double pv[4][4];
            for (int u = 0; u < 4; ++u) {
                const int jj = j + u < s.np ? j + u : s.np - 1;
                for (int t = 0; t < 4; ++t) pv[u][t] = c.M[s.o_pexp + jj * s.nq + tc[t]];
            }
            for (int u = 0; u < 4; ++u) {
                double acc = 0.0;
                for (int t = 0; t < 4; ++t) acc = fma(tv[t], pv[u][t], acc);
                if (pred && j + u < s.np) c.W[w_dst + (j + u) * s.nn + r] = acc;
            }
        }
    }
    wv::wave_fence();
}

// setlhs! (src/solvers.jl:46-96) in place on the n x n matrix at o_f (ROW-major, row pitch ld = GenHeader::ldf): partial
// pivoting (first strict maximum), full-row interchange, reciprocal on the diagonal; the interchanges composed into the
// gather at o_src.  Returns false for the instances that met an exactly zero pivot (the reference stops there: nothing of
// the factors is used afterwards).
// Row-major because the elimination's inner loop runs ALONG a row: a lane's entries (and the pivot row's) are contiguous,
// four of each come in with two ds_read2_b64, and nothing in the loop needs a predicate -- the last group of four may run
// up to three columns into the row's slack, where it updates numbers nobody reads.  (Column-major and predicated, the first
// version spent 12 instructions per multiply-add and 135 000 cycles per sample in here, at 20 unknowns.)
ACME_DEV bool coop_lu(const CoopCtx &c, int n, int o_f, int o_src) {
    double *W = c.W;
    const int ld = c.O.ld;
    for (int i = c.lig; i < n; i += c.lpi) W[o_src + i] = (double)i;
    wv::wave_fence();
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        double best = -1.0, bi = 1e9;
        for (int i = c.lig; i < n; i += c.lpi)
            if (i >= k) {
                const double v = fabs(W[o_f + i * ld + k]);
                if (v > best) { best = v; bi = (double)i; }
            }
        const double m = coop_allmax(c, best);
        // the reference starts from (amax = 0, kp = k) and takes the first strictly larger entry: the smallest index
        // holding the maximum; an all-zero (or all-NaN) column keeps kp = k
        double kpd = coop_allmin(c, (best == m && m > 0.0) ? bi : 1e9);
        const int kp = kpd < (double)n ? (int)kpd : k;
        const double piv = W[o_f + kp * ld + k];
        ok = ok && piv != 0.0;
        wv::wave_fence();
        if (kp != k) {
            for (int j = c.lig; j < n; j += c.lpi) {
                const double t = W[o_f + k * ld + j];
                W[o_f + k * ld + j] = W[o_f + kp * ld + j];
                W[o_f + kp * ld + j] = t;
            }
            if (c.lig == 0) {
                const double t = W[o_src + k];
                W[o_src + k] = W[o_src + kp];
                W[o_src + kp] = t;
            }
        }
        wv::wave_fence();
        const double inv = 1.0 / piv;
        const double *prow = W + o_f + k * ld + k + 1;              // the pivot row right of the diagonal
        const int cnt = n - k - 1;
        // (one row at a time: all of a lane's rows updated together -- pivot entries read once, every slot's operands
        // requested up front -- measured SLOWER, 135 000 against 96 000 cycles per sample: the predicated slots cost more
        // instructions than the shared reads save)
        for (int i = c.lig; i < n; i += c.lpi)
            if (i > k) {
                double *row = W + o_f + i * ld + k;
                const double l = row[0] * inv;
                row[0] = l;
                for (int j = 0; j < cnt; j += 4) {
                    double a0 = row[1 + j], a1 = row[2 + j], a2 = row[3 + j], a3 = row[4 + j];
                    const double b0 = prow[j], b1 = prow[j + 1], b2 = prow[j + 2], b3 = prow[j + 3];
                    a0 -= l * b0;
                    a1 -= l * b1;
                    a2 -= l * b2;
                    a3 -= l * b3;
                    row[1 + j] = a0;
                    row[2 + j] = a1;
                    row[3 + j] = a2;
                    row[4 + j] = a3;
                }
            }
        wv::wave_fence();
        if (c.lig == 0) W[o_f + k * ld + k] = inv;
        wv::wave_fence();
    }
    return ok;
}

// solve! (src/solvers.jl:98-132), x at w_x in place.  The right-hand side lives in REGISTERS during the two triangular
// sweeps (slot sl of lane l = row l + 16 sl); a step's x_j reaches the lanes through one ds_bpermute pair instead of an LDS
// write / read round trip: the sweeps are 2 n strictly sequential steps, and their latency is all they cost.  (The slot
// holding x_j is a compile-time index -- the sweeps are written out per slot: indexed by a run-time number the four
// registers became an array in scratch memory, a memory round trip per step.)
ACME_DEV void coop_lu_solve(const CoopCtx &c, int n, int o_f, int o_src, int w_x) {
    double *W = c.W;
    const int ld = c.O.ld;
    const int ns = (n + GROUP - 1) / GROUP;          // slots in use (uniform): the others' code is skipped, not predicated
    double xs[COOP_SLOTS];
    sfor<0, COOP_SLOTS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        xs[sl] = 0.0;
        if (sl < ns) {
            const int i = c.lig + c.lpi * sl;
            xs[sl] = i < n ? W[w_x + (int)W[o_src + i]] : 0.0;
        }
    });
    // forward: x_i -= F[i][j] x_j for i > j
    sfor<0, COOP_SLOTS>([&](auto sjc) ACME_LAMBDA {
        constexpr int sj = decltype(sjc)::value;
        if (sj < ns)
            for (int jj = 0; jj < GROUP && GROUP * sj + jj < n; ++jj) {
                const int j = GROUP * sj + jj;
                double f[COOP_SLOTS];
                sfor<sj, COOP_SLOTS>([&](auto sc) ACME_LAMBDA {          // (rows above slot sj are done)
                    constexpr int sl = decltype(sc)::value;
                    const int i = c.lig + c.lpi * sl;
                    f[sl] = (sl < ns && i > j && i < n) ? W[o_f + i * ld + j] : 0.0;
                });
                const double xj = wv::shfl16(xs[sj], jj);
                sfor<sj, COOP_SLOTS>([&](auto sc) ACME_LAMBDA {
                    constexpr int sl = decltype(sc)::value;
                    const int i = c.lig + c.lpi * sl;
                    if (sl < ns && i > j && i < n) xs[sl] -= f[sl] * xj;
                });
            }
    });
    // backward: x_j *= 1 / F[j][j] (stored), x_i -= F[i][j] x_j for i < j
    sfor_down<COOP_SLOTS>([&](auto sjc) ACME_LAMBDA {
        constexpr int sj = decltype(sjc)::value;
        if (sj < ns)
            for (int jj = GROUP - 1; jj >= 0; --jj) {
                const int j = GROUP * sj + jj;
                if (j >= n) continue;
                double f[COOP_SLOTS];
                sfor<0, sj + 1>([&](auto sc) ACME_LAMBDA {
                    constexpr int sl = decltype(sc)::value;
                    const int i = c.lig + c.lpi * sl;
                    f[sl] = i < j ? W[o_f + i * ld + j] : 0.0;
                });
                const double xj = W[o_f + j * ld + j] * wv::shfl16(xs[sj], jj);
                sfor<0, sj + 1>([&](auto sc) ACME_LAMBDA {
                    constexpr int sl = decltype(sc)::value;
                    const int i = c.lig + c.lpi * sl;
                    xs[sl] = i == j ? xj : (i < j ? xs[sl] - f[sl] * xj : xs[sl]);
                });
            }
    });
    wv::wave_fence();
    sfor<0, COOP_SLOTS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int i = c.lig + c.lpi * sl;
        if (sl < ns && i < n) W[w_x + i] = xs[sl];
    });
    wv::wave_fence();
}

// ---- the running factorisation in REGISTERS (17 ... 32 unknowns: kernels instantiated per NC = the unknowns rounded up to 4) ----
// The any-size path above walks evaluate! -> setlhs! -> solve! through LDS: every elimination step is a pivot search, an
// interchange and an update with LDS round trips a lone wave has nothing to hide behind (96 000 of 187 000 cycles per sample
// at 20 unknowns), and the two factor matrices are half of an instance's 16 KB -- 8 instances per compute unit, two waves,
// two of four SIMDs idle.  Here a lane computes its (up to two) Jacobian rows INTO registers, keeps them there through
// all n elimination steps (column indices are compile-time constants, the step loop is written out) and through the Newton
// step's triangular sweeps; only an ACCEPTED iterate's factors go to LDS (the extrapolation origin's matrix, the one
// factor matrix an instance keeps).  A row never moves: the reference's full-row interchange becomes the row's POSITION
// label (pos: where the row would sit after the interchanges so far) -- it decides a tie in the pivot search (first strict
// maximum = the smallest position), tells the sweeps their order, and says where the row is written on acceptance.  What
// crosses lanes goes through two small LDS buffers: a step's pivot row (written by its holder, read by everyone) and a
// sweep's x_j.  Arithmetic per entry: unchanged (l = a_ik * (1 / a_kk), a_ij -= l a_kj in the order k = 0, 1, ...;
// x_i -= F_ij x_j in the order of j), so factors, gather, zero-pivot verdict and solutions are the LDS path's bit for bit.
constexpr int COOP_REG_SLOTS = 2;

// evaluate!(nleq, z) as coop_evaluate, the rows' residuals, Jacobian rows (columns >= nn: zero) and Jq non-zeros in registers.
// rid: WHICH row of the sub-problem each slot of the lane evaluates (-1: none) -- the row the instance's learnt order puts
// at the slot's position (coop_gj_rows)
template <int NC>
ACME_DEV bool coop_evaluate_rows(const CoopCtx &c, const GenSub &s, int w_z, double (&a)[COOP_REG_SLOTS][NC],
                                 double (&res)[COOP_REG_SLOTS], double (&tvr)[COOP_REG_SLOTS][4], const int (&rid)[COOP_REG_SLOTS]) {
    constexpr int NS = COOP_REG_SLOTS;
    const GenHeader &H = c.H;
    for (int r = c.lig; r < s.nq; r += c.lpi) {
        c.W[c.O.q + r] = c.ell ? coop_ell_dot(c.M, s.e_fq, r, c.W + w_z, c.W[c.O.pf + r])
                                 : coop_dot(c.M + s.o_fq + r, s.nq, c.W + w_z, s.nn, c.W[c.O.pf + r]);
    }
    wv::wave_fence();
    bool bad = false;
    const wv::ExpTab etab = wv::load_exp_tab();
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int r = rid[sl];
        res[sl] = 0.0;
        for (int t = 0; t < 4; ++t) tvr[sl][t] = 0.0;
        sfor<0, NC>([&](auto jc) ACME_LAMBDA { a[sl][decltype(jc)::value] = 0.0; });
        if (r >= 0) {
            RowDesc rd;
            int tc[4];
            coop_rowdesc(c, s.row0 + r, rd, tc);
            double e[4], tv[4], rs;
            for (int t = 0; t < 4; ++t) e[t] = c.W[c.O.q + tc[t]];
            const bool expo = rd.kind == RK_DIODE || rd.kind == RK_BJT;
            const double exA = exp_junction(expo ? e[0] * rd.k[0] : 0.0, etab);
            const double exB = H.has_bjt ? exp_junction(rd.kind == RK_BJT ? e[1] * rd.k[1] : 0.0, etab) : 1.0;
            eval_row<true, 4>(rd, e, exA, exB, rs, tv);
            res[sl] = rs;
            bad = bad || !(rs * 0.0 == 0.0);
            for (int t = 0; t < 4; ++t) tvr[sl][t] = tv[t];
            sfor<0, NC / 4>([&](auto gc) ACME_LAMBDA {          // J row = Jq row * fq, four columns' operands at a time
                constexpr int j = 4 * decltype(gc)::value;
                double fv[4][4];
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u < s.nn ? j + u : s.nn - 1;
                    for (int t = 0; t < 4; ++t) fv[u][t] = c.M[s.o_fq + jj * s.nq + tc[t]];
                }
                sfor<0, 4>([&](auto uc) ACME_LAMBDA {
                    constexpr int u = decltype(uc)::value;
                    double acc = 0.0;
                    for (int t = 0; t < 4; ++t) acc = fma(tv[t], fv[u][t], acc);
                    const bool col = j + u < s.nn;
                    a[sl][j + u] = col ? acc : 0.0;
                    bad = bad || (col && !(acc * 0.0 == 0.0));
                });
            });
        }
    });
    return bad;
}

// calc_Jp closure with the Jq non-zeros of the latest coop_evaluate_rows (of the rows rid), written to the origin's Jp where
// `pred` -- row rid[sl] of Jp at row wrow[sl] of the stored matrix: the position the row holds, which is where the replayed
// elimination expects its right-hand side
ACME_DEV void coop_calc_jp_rows(const CoopCtx &c, const GenSub &s, const double (&tvr)[COOP_REG_SLOTS][4], bool pred,
                                const int (&rid)[COOP_REG_SLOTS], const int (&wrow)[COOP_REG_SLOTS]) {
    sfor<0, COOP_REG_SLOTS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int r = rid[sl];
        if (r >= 0) {
            const int blk = (s.row0 + r) / GROUP, ln = (s.row0 + r) % GROUP;
            int tc[4];
            for (int t = 0; t < 4; ++t) tc[t] = c.ti[blk * ROWI * GROUP + (3 + t) * GROUP + ln];
            for (int j = 0; j < s.np; j += 4) {
                double pv[4][4];
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u < s.np ? j + u : s.np - 1;
                    for (int t = 0; t < 4; ++t) pv[u][t] = c.M[s.o_pexp + jj * s.nq + tc[t]];
                }
                for (int u = 0; u < 4; ++u) {
                    double acc = 0.0;
                    for (int t = 0; t < 4; ++t) acc = fma(tvr[sl][t], pv[u][t], acc);
                    if (pred && j + u < s.np) c.W[c.O.ljp + (j + u) * s.nn + wrow[sl]] = acc;
                }
            }
        }
    });
    wv::wave_fence();
}

// setlhs! on the rows in registers; pos: the rows' positions in the factored matrix.  Returns false for the instances that
// met an exactly zero pivot.
// last (in / out): the positions the rows ended at in this instance's PREVIOUS factorisation (-1: none).  One Jacobian is
// much like the one before it, so the row that was the pivot of step k last time is nearly always the pivot again: it
// puts itself on the way to the others BEFORE the search has said so, and the search (two dependent reductions' worth of
// latency) then runs beside the LDS round trip instead of ahead of it.  The search still decides: unless it finds exactly
// that row -- the only lane of its instance holding the maximum, its smaller position on a tie inside the lane -- in all
// four instances of the wave, the step is done again the plain way (search, then hand-off).  Same pivots, same arithmetic.
template <int NC>
ACME_DEV bool coop_lu_rows(const CoopCtx &c, int n, double (&a)[COOP_REG_SLOTS][NC], int (&pos)[COOP_REG_SLOTS], int (&last)[COOP_REG_SLOTS]) {
    static_assert(NC % 4 == 0 && NC <= GROUP * COOP_REG_SLOTS, "columns in pairs, two rows per lane");
    constexpr int NS = COOP_REG_SLOTS;
    double *P = c.W + c.O.prow;               // the step's pivot row (16-byte aligned), behind it the position it came from
    int *Pk = reinterpret_cast<int *>(P + NC);
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {          // (a slot without a row: position -1, never a candidate, never updated)
        constexpr int sl = decltype(sc)::value;
        pos[sl] = c.lig + c.lpi * sl < n ? c.lig + c.lpi * sl : -1;
    });
    bool ok = true;
    sfor<0, NC>([&](auto kc) ACME_LAMBDA {
        constexpr int k = decltype(kc)::value;
        if (k < n) {
            // ---- last time's pivot row of this step sets out ----
            bool holds[NS];
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                holds[sl] = last[sl] == k && pos[sl] >= k;
                if (holds[sl]) {
                    sfor<k / 2, NC / 2>([&](auto gc) ACME_LAMBDA {
                        constexpr int g = decltype(gc)::value;
                        wv::st2(P + 2 * g, a[sl][2 * g], a[sl][2 * g + 1]);
                    });
                    Pk[0] = pos[sl];
                }
            });
            wv::lds_order();
            double b[NC];
            sfor<k / 2, NC / 2>([&](auto gc) ACME_LAMBDA {
                constexpr int g = decltype(gc)::value;
                const wv::pair_t v = wv::ld2(P + 2 * g);
                b[2 * g] = v.lo;
                b[2 * g + 1] = v.hi;
            });
            int kp = Pk[0];
            double inv = 1.0 / b[k];          // (beside the search, not behind its verdict: a dozen dependent operations)
            // ---- the pivot: the largest |a_ik| among the rows at positions >= k, the smallest position among equals ----
            double best = -1.0;
            int bp = 1 << 30;
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                const double v = fabs(a[sl][k]);
                const bool cand = pos[sl] >= k;
                if (cand && (v > best || (v == best && pos[sl] < bp))) {
                    best = v;
                    bp = pos[sl];
                }
            });
            const double m = wv::allmax16_nn(best);
            const bool top = best == m && m > 0.0;
            const bool sent = holds[0] || holds[1];
            const int sent_pos = holds[0] ? pos[0] : pos[1];
            const bool agree = m > 0.0 && (sent ? (top && bp == sent_pos) : !top);
            COOP_T(c, CT_LU_SEARCH);
            if (!ACME_USUAL(wv::ballot(!agree) == 0ull)) {
                // the plain way: who holds it (the second reduction, over the positions), then the hand-off
                const double kpd = wv::allmin16(top ? (double)bp : 1e9);
                kp = kpd < (double)n ? (int)kpd : k;
                sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                    constexpr int sl = decltype(sc)::value;
                    holds[sl] = pos[sl] == kp;
                    if (holds[sl])
                        sfor<k / 2, NC / 2>([&](auto gc) ACME_LAMBDA {
                            constexpr int g = decltype(gc)::value;
                            wv::st2(P + 2 * g, a[sl][2 * g], a[sl][2 * g + 1]);
                        });
                });
                wv::lds_order();
                sfor<k / 2, NC / 2>([&](auto gc) ACME_LAMBDA {
                    constexpr int g = decltype(gc)::value;
                    const wv::pair_t v = wv::ld2(P + 2 * g);
                    b[2 * g] = v.lo;
                    b[2 * g + 1] = v.hi;
                });
                wv::lds_order();          // (these reads before the next step's early write: program order on the GPU,
                                          //  a rendezvous for the emulator's lanes, which run one after the other)
                inv = 1.0 / b[k];
            }
            const double piv = b[k];
            ok = ok && piv != 0.0;
#ifdef ACME_COOP_TIMING
            inv = wv::keep(inv);
            COOP_T(c, CT_LU_HAND);
#endif
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                // (the interchange: positions k and kp trade places)
                pos[sl] = holds[sl] ? k : (pos[sl] == k ? kp : pos[sl]);
                if (pos[sl] > k) {
                    const double l = a[sl][k] * inv;
                    a[sl][k] = l;
                    sfor<k + 1, NC>([&](auto jc) ACME_LAMBDA {
                        constexpr int j = decltype(jc)::value;
                        a[sl][j] -= l * b[j];
                    });
                }
                if (pos[sl] == k) a[sl][k] = inv;          // the reciprocal on the diagonal (src/solvers.jl:86)
            });
        }
    });
    sfor<0, NS>([&](auto sc) ACME_LAMBDA { last[decltype(sc)::value] = pos[decltype(sc)::value]; });
    return ok;
}

// ---- the THRESHOLD path (17 ... 32 unknowns): the tuned 16-lane kernels' elimination, two rows per lane ----
// coop_lu_rows above reproduces the reference's dynamic first-strict-maximum pivoting bit for bit, and pays for it at every
// step: a search (16-lane reduction), the pivot row's round trip through LDS, an IEEE division -- 40 % of the kernel's
// cycles (profiles/r5_coop_timing.txt).  The tuned kernels (acme_kernel.h RowLU::solve_inplace) do not look for pivots:
// an instance keeps the row ORDER of the last factorisation that had to interchange rows -- position p (lane p mod 16,
// slot p / 16) evaluates row rid of the residual, so that the pivot of step k is where a compile-time DPP broadcast finds
// it -- and eliminates without a search, above and below the pivot (Gauss-Jordan: no triangular sweeps, the Newton step
// rides along as a column), with v_rcp_f64 + refinement for 1 / pivot and fused v_fmac_f64_dpp for "broadcast the pivot
// row's entry, multiply, accumulate".  The order is kept while every multiplier of a row still waiting to be the pivot
// satisfies |l| <= 8 (threshold partial pivoting, |pivot| >= max / 8: src/solvers.jl:58-78 is the case 1; the stated
// deviation of DESIGN a-13); an instance that trips the threshold -- or meets a zero pivot -- evaluates again, runs
// coop_lu_rows (the reference's pivoting) to LEARN the order, adopts it and eliminates again.  Either way the result is
// that of a valid LU of the same Jacobian with bounded growth: Newton steps agree with the reference's to rounding, the
// accepted iterate is decided by the residual test alone.
// What is kept of the factorisation at an extrapolation origin is the RECORDED elimination: column k of a row holds minus
// its multiplier of step k (0 in the pivot row itself), one more number its 1 / pivot; applied to another right-hand
// side (coop_replay_gj) it is the same linear map.  Unknown k comes out at position k whatever the row order was.
constexpr double COOP_PIVOT_THRESHOLD = 8.0;

// [A | rhs] -> A^-1 rhs (entry k at position k; dinv applied), a: the recorded elimination, dinv: the rows' 1 / pivot.
// Returns (per lane) whether its instance must not trust the result: threshold tripped, zero or non-finite pivot.
template <int NC>
ACME_DEV bool coop_gj_rows(const CoopCtx &c, int n, double (&a)[COOP_REG_SLOTS][NC], double (&rhs)[COOP_REG_SLOTS], double (&dinv)[COOP_REG_SLOTS]) {
    static_assert(COOP_REG_SLOTS == 2 && NC % 4 == 0 && NC <= 2 * GROUP, "two rows per lane");
    double vmx[2] = {0.0, 0.0}, frz[2] = {0.0, 0.0};
    dinv[0] = dinv[1] = 1.0;
    sfor<0, NC>([&](auto kc) ACME_LAMBDA {
        constexpr int k = decltype(kc)::value;
        constexpr int ks = k / GROUP, os = 1 - ks, kl = k % GROUP;
        if (k <= NC - 4 || k < n) {          // (uniform; only the last three steps of an instantiation can be beyond the model)
            wv::gj2_head<kl>(a[ks][k], a[os][k], dinv[ks], vmx[ks], vmx[os], frz[ks]);
            constexpr int M = NC - k;                           // the columns right of the pivot, and the right-hand side
            double *kp[M], *op[M];
            sfor<k + 1, NC>([&](auto jc) ACME_LAMBDA {
                constexpr int j = decltype(jc)::value;
                kp[j - k - 1] = &a[ks][j];
                op[j - k - 1] = &a[os][j];
            });
            kp[M - 1] = &rhs[ks];
            op[M - 1] = &rhs[os];
            wv::gj2_update<kl>(a[ks][k], a[os][k], kp, op);
        }
    });
    bool trip = false;
    sfor<0, 2>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        rhs[sl] *= dinv[sl];
        const bool real = c.lig + c.lpi * sl < n;
        // a zero pivot (exactly singular in this order) or a NaN turns the row's result into NaN, an infinite pivot leaves
        // 1 / pivot = 0 behind (RowLU::solve_inplace)
        trip = trip || (real && (frz[sl] > COOP_PIVOT_THRESHOLD || !(rhs[sl] * 0.0 == 0.0) || dinv[sl] == 0.0));
    });
    return coop_any(c, trip);
}

// an accepted iterate's recorded elimination into the origin's matrix in LDS (row p: position p's multipliers, behind them
// its 1 / pivot), for the instances with `pred`
template <int NC>
ACME_DEV void coop_store_gj(const CoopCtx &c, int n, const double (&a)[COOP_REG_SLOTS][NC], const double (&dinv)[COOP_REG_SLOTS], bool pred) {
    double *F = c.W + c.O.llu;
    sfor<0, COOP_REG_SLOTS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int p = c.lig + c.lpi * sl;
        if (pred && p < n) {
            sfor<0, NC / 2>([&](auto gc) ACME_LAMBDA {
                constexpr int g = decltype(gc)::value;
                wv::st2(F + p * c.O.ld + 2 * g, a[sl][2 * g], a[sl][2 * g + 1]);
            });
            F[p * c.O.ld + NC] = dinv[sl];
        }
    });
    wv::wave_fence();
}

// x <- A^-1 x with the elimination recorded at the origin (x at w_x, entry p the right-hand side of the row at position p
// on the way in, unknown p on the way out)
template <int NC> ACME_DEV void coop_replay_gj(const CoopCtx &c, int n, int w_x) {
    double *W = c.W;
    const double *F = W + c.O.llu;
    double m[2][NC], x[2], dinv[2];
    bool real[2];
    sfor<0, 2>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int p = c.lig + c.lpi * sl;
        real[sl] = p < n;
        x[sl] = real[sl] ? W[w_x + p] : 0.0;
        dinv[sl] = real[sl] ? F[p * c.O.ld + NC] : 0.0;
        sfor<0, NC / 2>([&](auto gc) ACME_LAMBDA {
            constexpr int g = decltype(gc)::value;
            wv::pair_t v{0.0, 0.0};
            if (real[sl]) v = wv::ld2(F + p * c.O.ld + 2 * g);
            m[sl][2 * g] = v.lo;
            m[sl][2 * g + 1] = v.hi;
        });
    });
    // (steps in runs of four: whole runs are inside the model for all but the last, and inside one slot)
    sfor<0, NC / 4>([&](auto gc) ACME_LAMBDA {
        constexpr int k0 = 4 * decltype(gc)::value;
        constexpr int ks = k0 / GROUP, os = 1 - ks;
        if (k0 + 4 < NC || k0 + 4 <= n) {
            wv::replay2_seg<k0, 4, k0>(x[ks], x[os], m[ks], m[os]);
        } else {
            sfor<0, 3>([&](auto uc) ACME_LAMBDA {
                constexpr int k = k0 + decltype(uc)::value;
                if (k < n) wv::replay2_seg<k, 1, k>(x[ks], x[os], m[ks], m[os]);
            });
        }
    });
    wv::wave_fence();
    sfor<0, 2>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        if (real[sl]) W[w_x + c.lig + c.lpi * sl] = x[sl] * dinv[sl];
    });
    wv::wave_fence();
}

// ---- the THRESHOLD path on a matrix in LDS (everything the register instantiations do not take: up to 64 unknowns) ----
// Beyond two rows per lane the Jacobian does not fit the registers (four rows of 64 columns are a wave's whole register
// file), and the literal LU in LDS pays a search, an interchange and a division at every step and two triangular sweeps per
// solve through LDS hand-offs (5.3e6 instance*samples/s at 34 unknowns).  Here the register instantiations' scheme -- the
// instance's learnt row order, no search while every multiplier has |l| <= 8, v_rcp_f64 + refinement, the reference's
// pivoting (coop_lu) only to learn a new order -- runs on ONE matrix per instance in LDS (instantiations NC = -NS, NS = the
// slots in use: rows per lane), as an LU factorisation that looks at what the matrix holds:
//   * position p (lane p mod 16, slot p / 16) owns row p of the matrix: evaluate! writes the Jacobian row of the residual
//     row the order puts there (rid) and its residual (column n: the right-hand side rides along).  Step k reads the pivot
//     row -- row k, whoever holds it: every lane of the instance reads the same addresses, an LDS broadcast -- and updates
//     the rows BELOW in place, a PAIR of columns (16 bytes) per read and write; column k of such a row becomes minus its
//     multiplier, 1 / pivot goes to a vector beside the matrix.  LU, not Gauss-Jordan as in registers: LDS stores are the
//     expensive operation here (a 16-byte store costs a wave 13 cycles and more, a read 4), and eliminating above the pivot
//     as well fills a banded matrix's upper triangle in.
//   * a circuit's Jacobian is sparse, and x + l * 0 = x exactly: pairs of columns in which the pivot rows of ALL the wave's
//     instances hold nothing are skipped, and so are the slots none of whose rows has a non-zero multiplier -- what remains
//     is the dense elimination's arithmetic, bit for bit.  The four pairs next to the pivot's and the right-hand side's are
//     read AHEAD of knowing what the pivot row holds (in a good order a circuit's entries sit near the diagonal), so a step
//     that needs nothing else is one round trip through LDS.
//   * the triangular sweeps keep x in registers and broadcast x_k by DPP (the step loop written out: compile-time lanes),
//     the factors read in runs of eight columns.
//   * the matrix an accepted iterate leaves behind IS the extrapolation origin's factorisation -- there is no second
//     matrix and no copy: an instance whose solve ends without an accepted iterate, or that rode along without needing one
//     (its matrix then holds some other linearisation), re-linearises at its origin before the origin is used again
//     (CoopSolver::fresh: the `reorig` pass of coop_simple_solve, which the launch start and solution-cache hits use anyway).
//     Half the LDS of the literal path: twice the resident instances.
template <int NS, int NP>
ACME_DEV void coop_lu_lds_chunk(double *const (&row)[NS], const bool (&act)[NS], bool real_last, const double *prow, const double (&m)[NS], int col) {
    wv::pair_t b[NP], a[NS][NP];
    sfor<0, NP>([&](auto uc) ACME_LAMBDA { constexpr int u = decltype(uc)::value; b[u] = wv::ld2(prow + col + 2 * u); });
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        if (act[sl])
            sfor<0, NP>([&](auto uc) ACME_LAMBDA { constexpr int u = decltype(uc)::value; a[sl][u] = wv::ld2(row[sl] + col + 2 * u); });
    });
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        if (act[sl] && (sl < NS - 1 || real_last))          // (only the last slot in use can hold positions beyond the matrix)
            sfor<0, NP>([&](auto uc) ACME_LAMBDA {
                constexpr int u = decltype(uc)::value;
                wv::st2(row[sl] + col + 2 * u, fma(m[sl], b[u].lo, a[sl][u].lo), fma(m[sl], b[u].hi, a[sl][u].hi));
            });
    });
}
// One step of the factorisation (ODD: k is odd -- column k is the upper half of its pair).  vmx: the lane's largest |multiplier|.
template <int NS, bool ODD, int LPI>
ACME_DEV void coop_lu_lds_step(const CoopCtx &c, int n, int k, double &vmx) {
    double *F = c.W + c.O.llu;
    const int ld = c.O.ld, kc = k & ~1;                   // (kc: the pair holding column k)
    const int ks = k / LPI;                               // (the slots below ks hold rows above the pivot: not touched)
    const double *prow = F + k * ld;
    const wv::pair_t pp = wv::ld2(prow + kc);
    double *row[NS];
    wv::pair_t own[NS];
    const bool real_last = c.lig + c.lpi * (NS - 1) < n;
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int p = c.lig + c.lpi * sl;
        row[sl] = F + ((sl < NS - 1 || real_last) ? p : n - 1) * ld;          // (a position beyond the matrix reads its last row, writes nothing)
        own[sl] = wv::pair_t{0.0, 0.0};
        if (sl >= ks) own[sl] = wv::ld2(row[sl] + kc);
    });
    // which of the pairs right of the pivot's hold anything in the pivot row -- of ANY of the wave's instances: lane l looks
    // at pairs l, l + 16 (, l + 32 beyond 62 columns: the right-hand side's pair of a 64-unknown system)
    const int g0 = kc / 2 + 1, g1 = n / 2;                // first and last pair of the update (g1: the right-hand side's)
    static_assert(LPI == GROUP, "one instance per wave has a step of its own: coop_lu_w64_step");
    unsigned long long todo = 0ull;
    sfor<0, 3>([&](auto hc) ACME_LAMBDA {
        constexpr int h = decltype(hc)::value;
        if (h < 2 || g1 >= 32) {
            const int g = c.lig + GROUP * h;
            const bool in = g >= g0 && g <= g1;
            const wv::pair_t v = wv::ld2(prow + 2 * (in ? g : g1));
            const unsigned long long bal = wv::ballot(in && !(v.lo == 0.0 && v.hi == 0.0));          // (a NaN counts as something)
            todo |= ((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull) << (GROUP * h);
        }
    });
    // read ahead: the four pairs next to the pivot's and the right-hand side's, of the pivot row and of the rows below
    constexpr int NB = 4;
    wv::pair_t bb[NB + 1], ab[NS][NB + 1];
    int gb[NB + 1];
    sfor<0, NB + 1>([&](auto uc) ACME_LAMBDA {
        constexpr int u = decltype(uc)::value;
        const int g = u < NB ? g0 + u : g1;
        gb[u] = g <= g1 ? g : g1;          // (beyond the row: the right-hand side's pair once more -- read, not written)
        bb[u] = wv::ld2(prow + 2 * gb[u]);
        sfor<0, NS>([&](auto sc) ACME_LAMBDA {
            constexpr int sl = decltype(sc)::value;
            if (sl >= ks) ab[sl][u] = wv::ld2(row[sl] + 2 * gb[u]);
        });
    });
    COOP_T(c, CT_S_SCAN);
    const double piv = ODD ? pp.hi : pp.lo;
    const double inv = wv::recip(piv);
    double m[NS];
    bool act[NS];
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int p = c.lig + c.lpi * sl;
        const bool below = sl >= ks && p > k;
        const double a = ODD ? own[sl].hi : own[sl].lo;
        m[sl] = below ? -a * inv : 0.0;
        vmx = fmax(vmx, fabs(m[sl]));
        act[sl] = sl >= ks && wv::ballot(m[sl] != 0.0) != 0ull;          // (a slot none of whose rows -- in any instance -- changes)
        if (act[sl] && (sl < NS - 1 || real_last)) {
            // column k of the rows below: minus the multiplier; the rows above (and the pivot's) keep what they hold
            if (ODD) wv::st2(row[sl] + kc, own[sl].lo, below ? m[sl] : own[sl].hi);
            else wv::st2(row[sl] + kc, below ? m[sl] : own[sl].lo, fma(m[sl], pp.hi, own[sl].hi));
        }
    });
    if (c.lig == k % LPI) c.W[c.O.dinv + k] = inv;
    COOP_T(c, CT_S_HEAD);
    // the pairs read ahead (the right-hand side's always -- on its own where it is not one of the four)
    sfor<0, NB + 1>([&](auto uc) ACME_LAMBDA {
        constexpr int u = decltype(uc)::value;
        const bool due = u < NB ? (g0 + u <= g1 && (((todo >> (g0 + u)) & 1ull) != 0ull || g0 + u == g1)) : g0 + NB <= g1;
        if (due)
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                if (act[sl] && (sl < NS - 1 || real_last))
                    wv::st2(row[sl] + 2 * gb[u], fma(m[sl], bb[u].lo, ab[sl][u].lo), fma(m[sl], bb[u].hi, ab[sl][u].hi));
            });
    });
    // what the pivot row holds beyond them: four pairs at a time where any of the four holds something, single pairs at the
    // row's end (short of the right-hand side's pair: that one is done)
    COOP_T(c, CT_S_BAND);
    int g = g0 + NB;
    for (; g + 4 <= g1; g += 4)
        if ((todo >> g) & 0xFull) {
            coop_lu_lds_chunk<NS, 4>(row, act, real_last, prow, m, 2 * g);
#ifdef ACME_COOP_TIMING
            c.tm->t[CT_S_CHUNKS] += 1;
#endif
        }
    for (; g < g1; ++g)
        if ((todo >> g) & 1ull) {
            coop_lu_lds_chunk<NS, 1>(row, act, real_last, prow, m, 2 * g);
#ifdef ACME_COOP_TIMING
            c.tm->t[CT_S_CHUNKS] += 1;
#endif
        }
#ifdef ACME_COOP_TIMING
    c.tm->t[CT_S_STEPS] += 1;
#endif
    COOP_T(c, CT_S_REST);
}
// The same step for ONE INSTANCE PER WAVE (one row per lane), as ONE round trip through LDS where the matrix keeps its
// pattern from one factorisation to the next -- it does: the pattern is the circuit's.  Beside the right-hand side's pair the
// step reads AHEAD the pairs of columns this step's pivot row held LAST time (pred: a 64-bit mask per step, lane k keeps
// step k's: CoopSolver::pm; the first two of them), of the pivot row and of the lane's own row -- and nothing else: with eight waves
// to a compute unit it is the LDS pipe the steps queue for (a 16-byte read of a wave takes it 4 cycles, a store 13), so a
// pair that holds nothing is not read on the off chance.  Every load of the step is requested before the first is waited
// for; whatever the pivot row holds that was not foreseen (a new row order, the first factorisation of a launch) takes the
// same way afterwards, two pairs to a round trip, and is foreseen next time (seen).
template <bool ODD>
ACME_DEV void coop_lu_w64_step(const CoopCtx &c, int n, int k, unsigned long long &big, double &dinv, unsigned long long pred, unsigned long long &seen) {
    double *F = c.W + c.O.llu;
    const int ld = c.O.ld, kc = k & ~1;
    const double *prow = F + k * ld;
    const bool real = c.lig < n;
    double *row = F + (real ? c.lig : n) * ld;          // (a lane beyond the matrix works on the spare row behind it: no predicates on the stores)
    const int g0 = kc / 2 + 1, g1 = n / 2;                // first and last pair of the update (g1: the right-hand side's)
    const bool rhs_apart = g0 <= g1;                      // (n odd, last step: the right-hand side shares the pivot's pair)
    const wv::pair_t pp = wv::ld2(prow + kc), own = wv::ld2(row + kc);
    const bool in = c.lig >= g0 && c.lig < g1;
    const wv::pair_t sv = wv::ld2(prow + 2 * (in ? c.lig : g1));          // lane l looks at pair l of the pivot row
    const wv::pair_t br = wv::ld2(prow + 2 * g1), ar = wv::ld2(row + 2 * g1);
#ifndef ACME_W64_NF
#define ACME_W64_NF 2          // (pairs read ahead per round trip: 1 / 2 / 3 / 4 / 6 / 8 measured, EXPERIMENTS.md -- a slot's mask arithmetic costs more than a round trip shared by the rare third pair saves)
#endif
    constexpr int NF = ACME_W64_NF;
    wv::pair_t bf[NF], af[NF];
    int gf[NF];
    // up to NF pairs of `from` requested (pivot row and own row); what was taken leaves `from`
    auto request = [&](unsigned long long &from) ACME_LAMBDA {
        sfor<0, NF>([&](auto ic) ACME_LAMBDA {
            constexpr int i = decltype(ic)::value;
            gf[i] = -1;
            if (from != 0ull) {
                gf[i] = __builtin_ctzll(from);
                from &= from - 1ull;
                bf[i] = wv::ld2(prow + 2 * gf[i]);
                af[i] = wv::ld2(row + 2 * gf[i]);
            }
        });
    };
    unsigned long long spec = pred;
    request(spec);
    COOP_T(c, CT_S_SCAN);
    unsigned long long rest = wv::ballot(in && !(sv.lo == 0.0 && sv.hi == 0.0));          // (a NaN counts as something)
    COOP_T(c, CT_S_HEAD);
    seen = rest;
    const double piv = ODD ? pp.hi : pp.lo;
    const double inv = wv::recip(piv);
    const bool below = real && c.lig > k;
    const double m = below ? -(ODD ? own.hi : own.lo) * inv : 0.0;
    big |= wv::ballot(fabs(m) > COOP_PIVOT_THRESHOLD);          // (the lanes whose multiplier trips the threshold, gathered over the steps)
    dinv = c.lig == k ? inv : dinv;                              // (lane k keeps 1 / pivot of step k: to LDS once, behind the last step)
#ifdef ACME_COOP_TIMING
    c.tm->t[CT_S_STEPS] += 1;
#endif
    COOP_T(c, CT_S_BAND);
    if (wv::ballot(m != 0.0) == 0ull) return;          // (no row below holds anything in column k)
    // column k of the rows below: minus the multiplier; the rows above (and the pivot's) keep what they hold
    if (ODD) wv::st2(row + kc, own.lo, below ? m : own.hi);
    else wv::st2(row + kc, below ? m : own.lo, fma(m, pp.hi, own.hi));
    if (rhs_apart) wv::st2(row + 2 * g1, fma(m, br.lo, ar.lo), fma(m, br.hi, ar.hi));
    // the requested pairs that the pivot row does hold are updated; `rest` keeps what is still to do
    auto apply = [&]() ACME_LAMBDA {
        sfor<0, NF>([&](auto ic) ACME_LAMBDA {
            constexpr int i = decltype(ic)::value;
            if (gf[i] >= 0 && ((rest >> gf[i]) & 1ull) != 0ull) {
                wv::st2(row + 2 * gf[i], fma(m, bf[i].lo, af[i].lo), fma(m, bf[i].hi, af[i].hi));
                rest &= ~(1ull << gf[i]);
            }
        });
    };
    apply();
    COOP_T(c, CT_S_REST);
    while (rest != 0ull) {          // what was not foreseen, or beyond the first NF: NF pairs to a round trip
#ifdef ACME_COOP_TIMING
        c.tm->t[CT_S_CHUNKS] += 1;
#endif
        unsigned long long more = rest;
        request(more);
        apply();
    }
}

// The triangular sweeps on x in registers (slot sl of the lane: position lig + 16 sl), the factors read from the lane's rows
// in runs of eight columns, x_k broadcast by DPP.
//   forward:  x_p += l_pk x_k  for p > k  (l: minus the multiplier, as stored), k ascending
template <int NS, int LPI> ACME_DEV void coop_lds_forward(int lig, int n, double (&x)[NS], const double *const (&row)[NS]) {
    sfor<0, NS * LPI / 8>([&](auto gc) ACME_LAMBDA {
        constexpr int k0 = 8 * decltype(gc)::value, ks = k0 / LPI;
        if (k0 < n) {
            double m[NS][8];
            sfor<ks, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                sfor<0, 4>([&](auto uc) ACME_LAMBDA {
                    constexpr int u = decltype(uc)::value;
                    const wv::pair_t v = wv::ld2(row[sl] + k0 + 2 * u);
                    m[sl][2 * u] = v.lo;
                    m[sl][2 * u + 1] = v.hi;
                });
            });
            sfor<0, 8>([&](auto uc) ACME_LAMBDA {
                constexpr int u = decltype(uc)::value, k = k0 + u, kl = k % LPI;
                if (k0 + 8 <= n || k < n) {          // (whole runs are inside the model for all but the last)
                    double xk;
                    if constexpr (LPI == 64) xk = wv::lane64<kl>(x[ks]);
                    else xk = wv::bcast16_ordered<kl, true>(x[ks]);
                    x[ks] = lig > kl ? fma(m[ks][u], xk, x[ks]) : x[ks];
                    sfor<ks + 1, NS>([&](auto sc) ACME_LAMBDA { constexpr int sl = decltype(sc)::value; x[sl] = fma(m[sl][u], xk, x[sl]); });
                }
            });
        }
    });
}
//   backward: x_k <- x_k / u_kk,  x_p -= u_pk x_k  for p < k,  k descending (the division: times the stored 1 / pivot; the
//   lanes' own x_p are scaled at the end, all at once -- x_p is not touched after its own step)
template <int NS, int LPI> ACME_DEV void coop_lds_backward(int lig, int n, double (&x)[NS], const double *const (&row)[NS], const double (&dinv)[NS]) {
    sfor_down<NS * LPI / 8>([&](auto gc) ACME_LAMBDA {
        constexpr int k0 = 8 * decltype(gc)::value, ks = k0 / LPI;
        if (k0 < n) {
            double m[NS][8];
            sfor<0, ks + 1>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                sfor<0, 4>([&](auto uc) ACME_LAMBDA {
                    constexpr int u = decltype(uc)::value;
                    const wv::pair_t v = wv::ld2(row[sl] + k0 + 2 * u);
                    m[sl][2 * u] = v.lo;
                    m[sl][2 * u + 1] = v.hi;
                });
            });
            sfor_down<8>([&](auto uc) ACME_LAMBDA {
                constexpr int u = decltype(uc)::value, k = k0 + u, kl = k % LPI;
                if (k0 + 8 <= n || k < n) {
                    double xk;
                    if constexpr (LPI == 64) xk = wv::lane64<kl>(x[ks] * dinv[ks]);
                    else xk = wv::bcast16_ordered<kl, true>(x[ks] * dinv[ks]);
                    x[ks] = lig < kl ? fma(-m[ks][u], xk, x[ks]) : x[ks];
                    sfor<0, ks>([&](auto sc) ACME_LAMBDA { constexpr int sl = decltype(sc)::value; x[sl] = fma(-m[sl][u], xk, x[sl]); });
                }
            });
        }
    });
    sfor<0, NS>([&](auto sc) ACME_LAMBDA { constexpr int sl = decltype(sc)::value; x[sl] *= dinv[sl]; });
}
// [J | res] of the matrix at O.llu -> its factorisation, x = J^-1 res (unknown p at position p).  Returns (per lane)
// whether its instance must not trust the result: threshold tripped, zero or non-finite pivot.
template <int NS, int LPI>
ACME_DEV bool coop_lu_lds(const CoopCtx &c, int n, double (&x)[NS], unsigned long long &pm) {
    double vmx = 0.0;
    if constexpr (LPI == 64) {
        unsigned long long big = 0ull;
        double dinv_k = 1.0;
        for (int k = 0; k < n; k += 2) {
            unsigned long long seen;
            coop_lu_w64_step<false>(c, n, k, big, dinv_k, wv::lanev64(pm, k), seen);
            pm = c.lig == k ? seen : pm;
            wv::lds_order();          // (a step reads what the step before wrote: the DS pipeline keeps a wave's program order)
            if (k + 1 < n) {
                coop_lu_w64_step<true>(c, n, k + 1, big, dinv_k, wv::lanev64(pm, k + 1), seen);
                pm = c.lig == k + 1 ? seen : pm;
                wv::lds_order();
            }
        }
        if (c.lig < n) c.W[c.O.dinv + c.lig] = dinv_k;
        wv::lds_order();
        vmx = big != 0ull ? 2.0 * COOP_PIVOT_THRESHOLD : 0.0;
    } else {
        for (int k = 0; k < n; k += 2) {
            coop_lu_lds_step<NS, false, LPI>(c, n, k, vmx);
            wv::lds_order();
            if (k + 1 < n) {
                coop_lu_lds_step<NS, true, LPI>(c, n, k + 1, vmx);
                wv::lds_order();
            }
        }
    }
#ifdef ACME_COOP_TIMING
    COOP_T(c, CT_LU_SEARCH);          // (developer timing: what the steps' own regions leave; the sweep under "setlhs!"; calls counted in the next slot)
    c.tm->t[CT_LU_HAND] += 1;
#endif
    const double *F = c.W + c.O.llu;
    const double *row[NS];
    double dinv[NS];
    bool real[NS];
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int p = c.lig + c.lpi * sl;
        real[sl] = p < n;
        row[sl] = F + (real[sl] ? p : n - 1) * c.O.ld;
        dinv[sl] = c.W[c.O.dinv + (real[sl] ? p : 0)];
        x[sl] = real[sl] ? row[sl][n] : 0.0;          // (the right-hand side came through the forward sweep as a column)
    });
    coop_lds_backward<NS, LPI>(c.lig, n, x, row, dinv);
    bool trip = vmx > COOP_PIVOT_THRESHOLD;
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        trip = trip || (real[sl] && (!(x[sl] * 0.0 == 0.0) || dinv[sl] == 0.0));
    });
    return coop_any(c, trip);
}

// evaluate!(nleq, z) as coop_evaluate, in the instance's row order: position p's row of the matrix <- the Jacobian row of
// residual row rid (columns 0 .. nn - 1) and that residual (column nn); the rows' Jq non-zeros to O.tv by position
template <int NS>
ACME_DEV bool coop_evaluate_lds(const CoopCtx &c, const GenSub &s, int w_z, const int (&rid)[COOP_SLOTS], double (&res)[NS]) {
    const GenHeader &H = c.H;
    double *F = c.W + c.O.llu;
    for (int r = c.lig; r < s.nq; r += c.lpi) {
        c.W[c.O.q + r] = c.ell ? coop_ell_dot(c.M, s.e_fq, r, c.W + w_z, c.W[c.O.pf + r])
                                 : coop_dot(c.M + s.o_fq + r, s.nq, c.W + w_z, s.nn, c.W[c.O.pf + r]);
    }
    wv::wave_fence();
    bool bad = false;
    const wv::ExpTab etab = wv::load_exp_tab();
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int r = rid[sl], p = c.lig + c.lpi * sl;
        res[sl] = 0.0;
        if (r >= 0) {
            RowDesc rd;
            int tc[4];
            coop_rowdesc(c, s.row0 + r, rd, tc);
            double e[4], tv[4], rs;
            for (int t = 0; t < 4; ++t) e[t] = c.W[c.O.q + tc[t]];
            const bool expo = rd.kind == RK_DIODE || rd.kind == RK_BJT;
            const double exA = exp_junction(expo ? e[0] * rd.k[0] : 0.0, etab);
            const double exB = H.has_bjt ? exp_junction(rd.kind == RK_BJT ? e[1] * rd.k[1] : 0.0, etab) : 1.0;
            eval_row<true, 4>(rd, e, exA, exB, rs, tv);
            res[sl] = rs;
            bad = bad || !(rs * 0.0 == 0.0);
            wv::st2(c.W + c.O.tv + 4 * p, tv[0], tv[1]);
            wv::st2(c.W + c.O.tv + 4 * p + 2, tv[2], tv[3]);
            double *row = F + p * c.O.ld;
            if (c.ell) {
                // J row = Jq row * fq from the row's sparse form (GenSub::o_jcol): zeros, then the columns that hold anything
                for (int j = 0; j < c.O.ld; j += 2) wv::st2(row + j, 0.0, 0.0);
#ifndef ACME_JROW_BATCH
#define ACME_JROW_BATCH 4          // (entries of a row's sparse form requested together)
#endif
                constexpr int JB = ACME_JROW_BATCH;
                for (int e = 0; e < s.kj; e += JB) {
                    int col[JB];
                    double cf[JB][4];
                    for (int u = 0; u < JB; ++u) {
                        const int ee = e + u < s.kj ? e + u : s.kj - 1;
                        col[u] = (int)c.M[s.o_jcol + ee * s.nn + r];
                        for (int t = 0; t < 4; ++t) cf[u][t] = c.M[s.o_jcoef + (ee * 4 + t) * s.nn + r];
                    }
                    for (int u = 0; u < JB; ++u) {
                        double acc = 0.0;
                        for (int t = 0; t < 4; ++t) acc = fma(tv[t], cf[u][t], acc);
                        bad = bad || !(acc * 0.0 == 0.0);
                        if (e + u < s.kj) row[col[u]] = acc;
                    }
                }
            } else
            for (int j = 0; j < s.nn; j += 4) {         // J row = Jq row * fq, four columns' operands at a time
                double fv[4][4], acc[4];
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u < s.nn ? j + u : s.nn - 1;
                    for (int t = 0; t < 4; ++t) fv[u][t] = c.M[s.o_fq + jj * s.nq + tc[t]];
                }
                for (int u = 0; u < 4; ++u) {
                    acc[u] = 0.0;
                    for (int t = 0; t < 4; ++t) acc[u] = fma(tv[t], fv[u][t], acc[u]);
                    acc[u] = j + u < s.nn ? acc[u] : 0.0;
                    bad = bad || !(acc[u] * 0.0 == 0.0);
                }
                wv::st2(row + j, acc[0], acc[1]);          // (the last group may reach into the row's slack: zeros)
                wv::st2(row + j + 2, acc[2], acc[3]);
            }
            row[s.nn] = rs;                                // (after the zeros of the last group)
        }
    });
    wv::wave_fence();
    return bad;
}

// calc_Jp closure with the Jq non-zeros coop_evaluate_lds left (where `pred`): Jp is kept by residual ROW here -- the pattern
// of a row's non-zeros is the model's, whatever position holds the row
template <int NS>
ACME_DEV void coop_calc_jp_lds(const CoopCtx &c, const GenSub &s, bool pred, const int (&rid)[COOP_SLOTS]) {
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int r = rid[sl], p = c.lig + c.lpi * sl;
        if (r >= 0) {
            const int blk = (s.row0 + r) / GROUP, ln = (s.row0 + r) % GROUP;
            int tc[4];
            double tv[4];
            for (int t = 0; t < 4; ++t) tc[t] = c.ti[blk * ROWI * GROUP + (3 + t) * GROUP + ln];
            for (int t = 0; t < 4; ++t) tv[t] = c.W[c.O.tv + 4 * p + t];
            if (c.ell) {
                for (int e = 0; e < s.kp; e += 4) {
                    int col[4];
                    double cf[4][4];
                    for (int u = 0; u < 4; ++u) {
                        const int ee = e + u < s.kp ? e + u : s.kp - 1;
                        col[u] = (int)c.M[s.o_pcol + ee * s.nn + r];
                        for (int t = 0; t < 4; ++t) cf[u][t] = c.M[s.o_pcoef + (ee * 4 + t) * s.nn + r];
                    }
                    for (int u = 0; u < 4; ++u) {
                        double acc = 0.0;
                        for (int t = 0; t < 4; ++t) acc = fma(tv[t], cf[u][t], acc);
                        if (pred && e + u < s.kp) c.W[c.O.ljp + (c.H.jp_sparse ? e + u : col[u]) * s.nn + r] = acc;
                    }
                }
            } else
            for (int j = 0; j < s.np; j += 4) {
                double pv[4][4];
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u < s.np ? j + u : s.np - 1;
                    for (int t = 0; t < 4; ++t) pv[u][t] = c.M[s.o_pexp + jj * s.nq + tc[t]];
                }
                for (int u = 0; u < 4; ++u) {
                    double acc = 0.0;
                    for (int t = 0; t < 4; ++t) acc = fma(tv[t], pv[u][t], acc);
                    if (pred && j + u < s.np) c.W[c.O.ljp + (j + u) * s.nn + r] = acc;
                }
            }
        }
    });
    wv::wave_fence();
}

// x <- A^-1 x with the factorisation in the matrix (x at w_x: entry p the right-hand side of the row at position p on the
// way in, unknown p on the way out)
template <int NS, int LPI> ACME_DEV void coop_replay_lds(const CoopCtx &c, int n, int w_x) {
    double *W = c.W;
    const double *F = W + c.O.llu;
    double x[NS], dinv[NS];
    const double *row[NS];
    bool real[NS];
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int p = c.lig + c.lpi * sl;
        real[sl] = p < n;
        row[sl] = F + (real[sl] ? p : n - 1) * c.O.ld;
        x[sl] = real[sl] ? W[w_x + p] : 0.0;
        dinv[sl] = W[c.O.dinv + (real[sl] ? p : 0)];
    });
    coop_lds_forward<NS, LPI>(c.lig, n, x, row);
    coop_lds_backward<NS, LPI>(c.lig, n, x, row, dinv);
    wv::wave_fence();
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        if (real[sl]) W[w_x + c.lig + c.lpi * sl] = x[sl];
    });
    wv::wave_fence();
}

// the extrapolation's solve! of a kernel instantiated for NC columns (0: the LDS version, any size)
template <int NC> ACME_DEV void coop_backsolve(const CoopCtx &c, int n, int o_f, int o_src, int w_x) {
    if constexpr (NC > 0) coop_replay_gj<NC>(c, n, w_x);
    else if constexpr (NC < 0) coop_replay_lds<coop_ns(NC), coop_lpi(NC)>(c, n, w_x);
    else coop_lu_solve(c, n, o_f, o_src, w_x);
}

// the solver of one instance's ONE sub-problem: where its current factors / its origin's factors sit (they trade places
// when an iterate is accepted: no copy of nn x nn doubles per sample)
struct CoopSolver {
    int o_lu, o_src;         // scratch of the running solve
    int o_llu, o_lsrc;       // the extrapolation origin's factors (last_linsolver, src/solvers.jl:191-196)
    int last[2];             // register instantiations: where this lane's rows ended in the latest factorisation (coop_lu_rows)
    int rid[2];              // ... which rows of the sub-problem the lane's two slots hold (-1: none).  Literal path: slot
                             // sl's own row lig + 16 sl, for good; threshold path: the instance's learnt row order
    bool fresh;              // threshold path: the origin (lp, lz) has no recorded elimination yet (launch start): the next
                             // solve linearises there first.  Matrix in LDS: ... or its one matrix holds something else by now
    int rid4[COOP_SLOTS];    // threshold path on the matrix in LDS: the row each of the lane's (up to four) positions holds
    unsigned long long pm;   // one instance per wave: lane k keeps which pairs of columns the pivot row of step k held last time
                             // (coop_lu_w64_step reads them ahead): a register, read with v_readlane -- as a word in LDS it cost
                             // every step a wait for the step before's stores
};
ACME_DEV void coop_accept_factors(CoopSolver &f, bool pred) {
    const int a = f.o_lu, b = f.o_src;
    f.o_lu = pred ? f.o_llu : f.o_lu;
    f.o_src = pred ? f.o_lsrc : f.o_src;
    f.o_llu = pred ? a : f.o_llu;
    f.o_lsrc = pred ? b : f.o_lsrc;
}

// Threshold path: evaluate!(nleq, z at w_z) and the elimination of [J | res] in the instance's current row order; where
// that order trips the pivot threshold (for the instances with `act`: nobody else's result is looked at) the rows are
// evaluated again, the reference's partially pivoted LU (coop_lu_rows) says where each row belongs, the instance adopts
// that order and everything is done once more.  ONE copy of evaluate! and of the elimination in the code: the three
// passes are trips of a loop.
//   out: a / dinv the recorded elimination, res = J^-1 res (unknown p at position p), tv the rows' Jq non-zeros,
//        resmax = max |res| before the solve (NaN: something was not finite), ok = false where the matrix is singular
//        (coop_lu_rows met an exactly zero pivot: src/solvers.jl:80-83) -- both per instance
template <int NC>
ACME_DEV void coop_linearize(const CoopCtx &c, const GenSub &s, CoopSolver &f, int w_z, bool act, double (&a)[COOP_REG_SLOTS][NC],
                             double (&res)[COOP_REG_SLOTS], double (&tv)[COOP_REG_SLOTS][4], double (&dinv)[COOP_REG_SLOTS],
                             double &resmax, bool &ok) {
    constexpr int NS = COOP_REG_SLOTS;
    const int nn = s.nn;
    int phase = 0;          // 0: the usual pass; 1: learning the order; 2: once more in the new order
    bool learn = false;     // this instance is adopting a new order
    ok = true;
    for (;;) {
        const bool bad = coop_evaluate_rows<NC>(c, s, w_z, a, res, tv, f.rid);
        COOP_T(c, CT_EVAL);
        if (phase == 1) {
            int pos[NS];
            const bool okl = coop_lu_rows<NC>(c, nn, a, pos, f.last);
            ok = learn ? okl : ok;
            // the row that ended at position p moves to position p (through the x_j hand-off buffer, as integers)
            int *ob = reinterpret_cast<int *>(c.W + c.O.xb);
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                if (f.rid[sl] >= 0) ob[pos[sl]] = f.rid[sl];
            });
            wv::wave_fence();
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                const int p = c.lig + c.lpi * sl;
                if (learn && p < nn) {
                    f.rid[sl] = ob[p];
                    f.last[sl] = p;          // (where this row would end if the same matrix were factored again)
                }
            });
            wv::wave_fence();
            COOP_T(c, CT_LU);
            phase = 2;
            continue;
        }
        const bool finite = !coop_any(c, bad);
        double rm = 0.0;
        sfor<0, NS>([&](auto sc) ACME_LAMBDA {
            constexpr int sl = decltype(sc)::value;
            const double v = fabs(res[sl]);
            if (f.rid[sl] >= 0 && v > rm) rm = v;
        });
        resmax = wv::allmax16(rm);
        if (!finite) resmax = (double)NAN;
        const bool trip = coop_gj_rows<NC>(c, nn, a, res, dinv);
        COOP_T(c, CT_LU);
        if (phase == 0) {
            learn = act && finite && trip;
            if (ACME_USUAL(wv::ballot(learn) == 0ull)) break;
            ACME_DBG("coop relearn: instance %lld lane %d learn %d", c.i, c.lig, (int)learn);
            phase = 1;
            continue;
        }
        ok = ok && !(learn && trip);          // (cannot happen behind partial pivoting short of a singular matrix)
        break;
    }
}

// coop_linearize on the matrix in LDS: evaluate!(nleq, z at w_z) into the instance's one matrix, the elimination in place;
// x = J^-1 res (unknown p at position p).  Learning a new order: the reference's pivoting on the same matrix (coop_lu: real
// interchanges, the gather src[] says which of the present positions' rows ends where), adopted, and everything once more.
template <int NS, int LPI>
ACME_DEV void coop_linearize_lds(const CoopCtx &c, const GenSub &s, CoopSolver &f, int w_z, bool act, double (&x)[NS], double &resmax, bool &ok) {
    const int nn = s.nn;
    int phase = 0;
    bool learn = false;
    ok = true;
    for (;;) {
        double res[NS];
        const bool bad = coop_evaluate_lds<NS>(c, s, w_z, f.rid4, res);
        COOP_T(c, CT_EVAL);
        if (phase == 1) {
            const bool okl = coop_lu(c, nn, c.O.llu, c.O.lsrc);
            ok = learn ? okl : ok;
            int *ob = reinterpret_cast<int *>(c.W + c.O.xb);
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                if (f.rid4[sl] >= 0) ob[c.lig + c.lpi * sl] = f.rid4[sl];
            });
            wv::wave_fence();
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                const int p = c.lig + c.lpi * sl;
                if (learn && p < nn) f.rid4[sl] = ob[(int)c.W[c.O.lsrc + p]];
            });
            wv::wave_fence();
            COOP_T(c, CT_LU);
            phase = 2;
            continue;
        }
        const bool finite = !coop_any(c, bad);
        double rm = 0.0;
        sfor<0, NS>([&](auto sc) ACME_LAMBDA {
            constexpr int sl = decltype(sc)::value;
            const double v = fabs(res[sl]);
            if (f.rid4[sl] >= 0 && v > rm) rm = v;
        });
        resmax = coop_allmax(c, rm);
        if (!finite) resmax = (double)NAN;
        const bool trip = coop_lu_lds<NS, LPI>(c, nn, x, f.pm);
        COOP_T(c, CT_LU);
        if (phase == 0) {
            learn = act && finite && trip;
            if (ACME_USUAL(wv::ballot(learn) == 0ull)) break;
            ACME_DBG("coop relearn (LDS): instance %lld lane %d learn %d", c.i, c.lig, (int)learn);
            phase = 1;
            continue;
        }
        ok = ok && !(learn && trip);
        break;
    }
}

// set_extrapolation_origin(solver, p, z) (src/solvers.jl:183-196) at (w_lp, w_lz) for the instances with `pred` (the any-size
// instantiation; the register instantiations re-linearise inside coop_simple_solve: `reorig`, one copy of the pass)
ACME_DEV void coop_set_origin(const CoopCtx &c, const GenSub &s, CoopSolver &f, bool pred) {
    coop_set_p(c, s, c.O.lp);
    (void)coop_evaluate(c, s, c.O.lz, f.o_lu);
    (void)coop_lu(c, s.nn, f.o_lu, f.o_src);
    coop_calc_jp(c, s, c.O.ljp, pred);
    coop_accept_factors(f, pred);
}

// solve(::SimpleSolver, p) (src/solvers.jl:207-236) for the instances with `need`: p at w_p, z left in w_zz; returns
// hasconverged, needediterations in its.
// reorig (threshold path only): set_extrapolation_origin(solver, last_p, last_z) (src/solvers.jl:183-196) first, for these
// instances -- their origin was replaced by a stored solution (coop_cached_solve) or has no recorded elimination yet (launch
// start).  It is a pass of the SAME loop as the Newton iterations: evaluate!, the elimination and the re-learning exist
// once in the kernel (what Shape::ONELOOP is to the tuned kernels), not once per caller.
template <int NC> ACME_DEV bool coop_simple_solve(const CoopCtx &c, const GenSub &s, CoopSolver &f, int w_p, bool need, int &its, bool reorig = false) {
    const GenHeader &H = c.H;
    double *W = c.W;
    const int nn = s.nn, np = s.np;
    COOP_T(c, CT_REST);
    // the start: z <- last_z - last_J \ (last_Jp (p - last_p))  (src/solvers.jl:209-215)
    auto start = [&]() ACME_LAMBDA {
        coop_set_p(c, s, w_p);
        COOP_T(c, CT_SETP);
        if constexpr (NC < 0) {
            // (Jp by residual row: position p takes the row it holds; from the row's sparse form where the image has it)
            for (int sl = 0; sl < coop_ns(NC); ++sl) {
                const int r = f.rid4[sl];
                if (r < 0) continue;
                double acc = 0.0;
                if (c.ell) {
                    for (int e = 0; e < s.kp; e += 4) {
                        double jv[4], dv[4];
                        for (int u = 0; u < 4; ++u) {
                            const int ee = e + u < s.kp ? e + u : s.kp - 1;
                            const int col = (int)c.M[s.o_pcol + ee * nn + r];
                            jv[u] = W[c.O.ljp + (H.jp_sparse ? ee : col) * nn + r];
                            dv[u] = W[w_p + col] - W[c.O.lp + col];
                        }
                        for (int u = 0; u < 4; ++u)
                            if (e + u < s.kp) acc = fma(jv[u], dv[u], acc);
                    }
                } else {
                    acc = coop_dot_diff(W + c.O.ljp + r, nn, W + w_p, W + c.O.lp, np, 0.0);
                }
                W[c.O.tmp + c.lig + c.lpi * sl] = acc;
            }
        } else {
            for (int r = c.lig; r < nn; r += c.lpi) W[c.O.tmp + r] = coop_dot_diff(W + c.O.ljp + r, nn, W + w_p, W + c.O.lp, np, 0.0);
        }
        wv::wave_fence();
        coop_backsolve<NC>(c, nn, f.o_llu, f.o_lsrc, c.O.tmp);
        for (int r = c.lig; r < nn; r += c.lpi)
            if (need) W[c.O.zz + r] = W[c.O.lz + r] - W[c.O.tmp + r];
        wv::wave_fence();
        COOP_T(c, CT_EXTRAP);
    };
    bool act = need, conv = false;
    double reslast = 0.0;
    its = 0;
    if constexpr (NC != 0) {
        constexpr int NS = NC > 0 ? COOP_REG_SLOTS : coop_ns(NC);
        int stage = wv::ballot(reorig) != 0ull ? 0 : 1;          // 0: re-linearising at the origin; 1: the start is due; 2: Newton
        for (;;) {
            if (stage == 0) {
                coop_set_p(c, s, c.O.lp);
            } else {
                if (stage == 1) {
                    start();
                    stage = 2;
                }
                if (wv::ballot(act) == 0ull) break;
                its += act ? 1 : 0;
            }
            const bool origin = stage == 0;
            constexpr int NCR = NC > 0 ? NC : 1;          // (the registers' arrays: not used by the instantiations on a matrix in LDS)
            double a[COOP_REG_SLOTS][NCR], res[NS], tv[COOP_REG_SLOTS][4], dinv[COOP_REG_SLOTS], resmax;
            bool ok;
            if constexpr (NC > 0) coop_linearize<NC>(c, s, f, origin ? c.O.lz : c.O.zz, origin ? reorig : act, a, res, tv, dinv, resmax, ok);
            else coop_linearize_lds<NS, coop_lpi(NC)>(c, s, f, origin ? c.O.lz : c.O.zz, origin ? reorig : act, res, resmax, ok);
            const bool finite = resmax == resmax;
            const bool small = resmax < c.A.tol;
            const bool accept = !origin && act && finite && ok && small;
            const bool step = !origin && act && finite && ok && !small;
            reslast = (!origin && act) ? resmax : reslast;
            // the Newton step came out of the elimination: unknown p at position p
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                const int p = c.lig + c.lpi * sl;
                if (step && p < nn) W[c.O.zz + p] -= res[sl];
            });
            COOP_T(c, CT_SOLVE);
            // an accepted iterate: its recorded elimination, Jp, p and z become the extrapolation origin; an origin that
            // is being re-linearised keeps its p and z
            const bool rec = origin ? reorig : accept;
            if (wv::ballot(rec) != 0ull) {
                if constexpr (NC > 0) {
                    const int wrow[COOP_REG_SLOTS] = {c.lig, c.lig + GROUP};
                    coop_calc_jp_rows(c, s, tv, rec, f.rid, wrow);
                    coop_store_gj<NC>(c, nn, a, dinv, rec);
                } else {
                    coop_calc_jp_lds<NS>(c, s, rec, f.rid4);          // (the matrix is the record already)
                }
                for (int j = c.lig; j < np; j += c.lpi)
                    if (accept) W[c.O.lp + j] = W[w_p + j];
                for (int r = c.lig; r < nn; r += c.lpi)
                    if (accept) W[c.O.lz + r] = W[c.O.zz + r];
            }
            wv::wave_fence();
            COOP_T(c, CT_ACCEPT);
            if (origin) {
                stage = 1;
                continue;
            }
            conv = conv || accept;
            act = step && its < c.A.maxiter;
        }
        // matrix in LDS: it is the origin's recorded elimination only where this solve ended with an accepted iterate
        if constexpr (NC < 0) f.fresh = !(need && conv);
    } else {
        start();
        while (wv::ballot(act) != 0ull) {
            its += act ? 1 : 0;
            const bool bad = coop_evaluate(c, s, c.O.zz, f.o_lu);
            COOP_T(c, CT_EVAL);
            const bool finite = !coop_any(c, bad);
            double rm = 0.0;
            for (int r = c.lig; r < nn; r += c.lpi) {
                const double v = fabs(W[c.O.res + r]);
                if (v > rm) rm = v;
            }
            double resmax = wv::allmax16(rm);
            if (!finite) resmax = (double)NAN;
            const bool ok = coop_lu(c, nn, f.o_lu, f.o_src);
            COOP_T(c, CT_LU);
            const bool small = resmax < c.A.tol;
            const bool accept = act && finite && ok && small;
            const bool step = act && finite && ok && !small;
            reslast = act ? resmax : reslast;
            // the Newton step (for everyone; only the stepping instances keep it)
            for (int r = c.lig; r < nn; r += c.lpi) W[c.O.dz + r] = W[c.O.res + r];
            wv::wave_fence();
            coop_backsolve<NC>(c, nn, f.o_lu, f.o_src, c.O.dz);
            for (int r = c.lig; r < nn; r += c.lpi)
                if (step) W[c.O.zz + r] -= W[c.O.dz + r];
            COOP_T(c, CT_SOLVE);
            // an accepted iterate: its factors, Jp, p and z become the extrapolation origin
            if (wv::ballot(accept) != 0ull) {
                coop_calc_jp(c, s, c.O.ljp, accept);
                coop_accept_factors(f, accept);
                for (int j = c.lig; j < np; j += c.lpi)
                    if (accept) W[c.O.lp + j] = W[w_p + j];
                for (int r = c.lig; r < nn; r += c.lpi)
                    if (accept) W[c.O.lz + r] = W[c.O.zz + r];
            }
            wv::wave_fence();
            COOP_T(c, CT_ACCEPT);
            conv = conv || accept;
            act = step && its < c.A.maxiter;
        }
    }
    (void)conv;
    return reslast < c.A.tol;          // hasconverged (:203): false for a NaN residual
}

// solve(::CachingSolver, p) (src/solvers.jl:347-396) with the bounded store (lane e looks at stored solution e)
template <int NC> ACME_DEV bool coop_cached_solve(const CoopCtx &c, const GenSub &s, CoopSolver &f, int w_p, bool need, int &its) {
    double *W = c.W;
    const int nn = s.nn, np = s.np;
    double *cp = c.Cp;                                                                       // LDS (coop_main loads / stores it)
    int *meta = reinterpret_cast<int *>(cp + np * CACHE);
    double *cz = c.A.cache + (c.valid ? c.i : 0) * c.H.cache_total + s.c_off + np * CACHE + 2;   // the stored z's: HBM
    const bool caching = c.A.solver == SOLVER_CACHING_HOMOTOPY;
    // (threshold path: the origin still lacks its recorded elimination -- launch start; matrix in LDS: or has lost it,
    // which matters once the instance needs a solve again)
    bool reorig = NC < 0 ? f.fresh && need : f.fresh;
    f.fresh = NC < 0 ? f.fresh && !need : false;
    if (caching) {
        static_assert(CACHE == GROUP, "one stored solution per lane");
        double best = 0.0, d = 0.0;
        const int count = c.valid ? meta[0] : 0;
        for (int j = 0; j < np; ++j) {
            const double pj = W[w_p + j];
            const double dl = pj - W[c.O.lp + j];
            best = fma(dl, dl, best);
            const double t = (c.valid ? cp[j * CACHE + (c.lig & (CACHE - 1))] : 0.0) - pj;
            d = fma(t, t, d);
        }
        d = c.lig < count ? d : (double)INFINITY;
        const double m = coop_allmin(c, d);
        const unsigned long long bal = wv::ballot(d == m);
        const int idx = wv::ffs32((int)((bal >> (c.grp * GROUP)) & 0xFFFFull)) - 1;
        const bool hit = need && count > 0 && m < best;
        if (wv::ballot(hit) != 0ull) {
            const int e = hit ? idx : 0;
            for (int j = c.lig; j < np; j += c.lpi)
                if (hit) W[c.O.lp + j] = cp[j * CACHE + e];
            for (int r = c.lig; r < nn; r += c.lpi)
                if (hit) W[c.O.lz + r] = cz[e * nn + r];
            wv::wave_fence();
            if constexpr (NC != 0) reorig = reorig || hit;         // (a pass of coop_simple_solve's loop)
            else coop_set_origin(c, s, f, hit);
        }
    }
    COOP_T(c, CT_LOOKUP);
    const bool conv = coop_simple_solve<NC>(c, s, f, w_p, need, its, reorig);
    if (caching) {
        const bool keep = need && conv && its > 5;
        if (wv::ballot(keep) != 0ull) {
            const int count = c.valid ? meta[0] : 0, head = c.valid ? meta[1] : 0;
            const int slot = count < CACHE ? count : head;
            wv::wave_fence();
            for (int j = c.lig; j < np; j += c.lpi)
                if (keep) cp[j * CACHE + slot] = W[w_p + j];
            for (int r = c.lig; r < nn; r += c.lpi)
                if (keep && c.wr) cz[slot * nn + r] = W[c.O.zz + r];
            if (keep && c.lig == 0) {
                meta[0] = count < CACHE ? count + 1 : count;
                meta[1] = count < CACHE ? head : (head + 1) & (CACHE - 1);
            }
            wv::wave_fence();
        }
    }
    return conv;
}

// solve(::HomotopySolver, p) (src/solvers.jl:268-296); p at w_p of the header.  ONE loop whose first pass is the direct
// attempt (at w_p) and whose later passes are the bisection's (at w_pa): one inlined copy of the solver stack in the kernel
// instead of two (what Shape::ONELOOP is to the tuned kernels: half the code for the instruction cache to hold).
template <int NC> ACME_DEV bool coop_homotopy_solve(const CoopCtx &c, const GenSub &s, CoopSolver &f, bool need0, int &its_total) {
    const GenHeader &H = c.H;
    double *W = c.W;
    bool conv = false, need = need0, direct = true;
    double a = 0.5, best = 0.0;
    int w_src = c.O.p;
    its_total = 0;
    do {
        int its;
        const bool cv = coop_cached_solve<NC>(c, s, f, w_src, need, its);
        its_total += need ? its : 0;
        conv = need ? cv : conv;
        if (direct) {
            // the direct attempt failed for these: the homotopy starts from the origin it left behind
            need = need && !cv && c.A.solver != SOLVER_SIMPLE;
            direct = false;
            if (wv::ballot(need) != 0ull) {
                for (int j = c.lig; j < s.np; j += c.lpi)
                    if (need) W[c.O.sp + j] = W[c.O.lp + j];
                wv::wave_fence();
            }
        } else {
            if (need) {
                if (cv) {
                    best = a;
                    a = 1.0;
                } else {
                    const double na = (a + best) / 2.0;
                    if (!(best < na && na < a)) need = false;
                    a = na;
                }
            }
            need = need && best < 1.0;
        }
        if (wv::ballot(need) == 0ull) break;
        for (int j = c.lig; j < s.np; j += c.lpi) {
            double pa = W[c.O.sp + j] * (1.0 - a);
            pa = pa + a * W[c.O.p + j];
            if (need) W[c.O.pa + j] = pa;
        }
        wv::wave_fence();
        w_src = c.O.pa;
    } while (true);
    return conv;
}

// run! for the instances of one wave (GArgs::mode == GEN_RUN).  lds: the BLOCK's LDS -- GArgs::coop_wpb waves share one
// copy of the row tables and (IMGL: the batch shares one model image) of the image, staged by all of them; behind that every
// wave has its instances' workspaces.  A dependent load from L2 costs a lone wave ~1 us, and evaluate! alone chains five of
// them: with four waves to a block the image of a 20-unknown model fits beside 16 instances' workspaces.  After the one
// barrier behind the staging a wave never talks to another.
template <bool IMGL, int NC> ACME_DEV void coop_main(const GArgs &A, double *lds, int wave_in_block, int wave_global, int lane) {
    const GenHeader &H = *A.H;
    constexpr int LPI = coop_lpi(NC);          // lanes per instance
    const int lig = lane & (LPI - 1), grp = lane / LPI;
    const int gpw = A.coop_gpw, wpb = A.coop_wpb;
    lds = static_cast<double *>(__builtin_assume_aligned(lds, 16));
    // ---- the block's shared part: model image (if shared) and row tables, loaded by all its lanes ----
    double *img = lds;
    const int img0 = coop_image_first(H, NC);          // (the part of the image this instantiation reads)
    double *tk = lds + (IMGL ? coop_image_doubles(H, NC) : 0);
    const int blocks = (H.nnt + GROUP - 1) / GROUP;
    int *ti = reinterpret_cast<int *>(tk + blocks * 8 * GROUP);
    const int t0 = wave_in_block * 64 + lane, tstep = wpb * 64;
    if constexpr (IMGL)
        for (int k = t0; k < coop_image_doubles(H, NC) && img0 + k < H.image_total; k += tstep) img[k] = A.image[img0 + k];
    for (int k = t0; k < blocks * 8 * GROUP; k += tstep) {
        const int blk = k / (8 * GROUP), rest = k % (8 * GROUP);
        tk[k] = A.rowc[(long long)blk * ROWC * GROUP + rest];       // constants 0 .. 7 of the block's 16 rows
    }
    for (int k = t0; k < blocks * ROWI * GROUP; k += tstep) ti[k] = A.rowi[k];
    if (wpb > 1) wv::block_sync();
    else wv::wave_fence();
    const long long slot = (long long)wave_global * gpw + grp;
    // A row of 16 lanes WITHOUT an instance of its own (a wave carrying fewer than four; the last wave of a batch) does not
    // leave: it rides along as a MIRROR of the wave's first row -- the same instance, the same workspace in LDS, hence the
    // same values in every register and the same (redundant) LDS writes -- with its writes to global memory switched off
    // (`wr`).  The wave then runs with all 64 lanes from the first instruction to the last.  Leaving early (round 5) put the
    // whole solver under a partial EXEC mask, and the any-size instantiation, built without its register bound, then
    // computed zeros whenever a wave carried fewer than four instances (tools/zeros_probe.py): with every lane
    // active that build is right, whatever the register allocation.  (A wave whose FIRST row has no instance has none at all
    // and leaves as a whole.)
    const bool wr = grp < gpw && slot < A.n_inst;
    if ((long long)wave_global * gpw >= A.n_inst) return;          // (a surplus wave of the last block: all 64 lanes leave)
    // (the CPU wave emulator of the tests runs the lanes of a wave one after the other between rendezvous: a mirror would
    // apply every LDS read-modify-write a second time there -- its rows leave instead, it has no EXEC mask to get wrong)
    if (!wr && !wv::lockstep()) return;
    const bool valid = true;
    const long long i = wr ? slot : (long long)wave_global * gpw;
    const CoopOff O = coop_offsets(H, NC);
    double *W = lds + coop_shared_doubles(H, IMGL, NC) + (long long)(wave_in_block * gpw + (wr ? grp : 0)) * coop_inst_doubles(H, NC);
    double *Cp = W + ((O.total + 1) & ~1);
    CoopCtx c{A, H, O, IMGL ? img - img0 : A.image + i * A.image_stride, W, Cp, tk, ti, lig, grp, LPI, coop_reads_sparse(H, NC), i, valid, wr};
#ifdef ACME_COOP_TIMING
    CoopTimer tmr{};
    tmr.mark = (long long)__builtin_readcyclecounter();
    c.tm = &tmr;
#endif
    double *st = A.state + i * H.state_total;
    long long *rep = A.report + i * RW_WORDS;
    const bool has_sub = H.nsub > 0;
    const GenSub &s = H.sub[0];
    CoopSolver f{c.O.lu, c.O.src, has_sub ? c.O.llu : 0, has_sub ? c.O.lsrc : 0, {-1, -1}, {-1, -1}, NC != 0, {-1, -1, -1, -1}};
    // the rows the lane's two slots hold: what the instance's order -- learnt in earlier launches, GArgs::coop_order -- puts
    // at the slots' positions
    if constexpr (NC > 0)
        for (int sl = 0; sl < 2; ++sl) {
            const int p = lig + LPI * sl;
            f.rid[sl] = (has_sub && p < s.nn) ? A.coop_order[i * COOP_MAX_N + p] : -1;
        }
    if constexpr (NC < 0)
        for (int sl = 0; sl < coop_ns(NC); ++sl) {
            const int p = lig + LPI * sl;
            f.rid4[sl] = (has_sub && p < s.nn) ? A.coop_order[i * COOP_MAX_N + p] : -1;
        }
    const bool caching = has_sub && A.solver == SOLVER_CACHING_HOMOTOPY;
    double *cache_g = A.cache + i * H.cache_total + (has_sub ? s.c_off : 0);
    for (int k = lig; k < O.total; k += LPI) W[k] = 0.0;
    wv::wave_fence();
    for (int k = lig; k < H.nx; k += LPI) W[c.O.x + k] = st[k];
    if (has_sub) {
        for (int j = lig; j < s.np; j += LPI) W[c.O.lp + j] = st[H.nx + s.poff + j];
        for (int r = lig; r < s.nn; r += LPI) W[c.O.lz + r] = st[H.nx + H.npt + s.zoff + r];
        if (caching)       // the stored p's and the two counters live in LDS for the launch
            for (int k = lig; k < s.np * CACHE + 2; k += LPI) Cp[k] = cache_g[k];
    }
    wv::wave_fence();
    if constexpr (NC == 0)
        if (has_sub) coop_set_origin(c, s, f, true);
    bool dead = rep[RW_FIRST_NONFINITE] >= 0;
    long long it_total = 0, it_max = 0;
    // this sample's inputs sit in LDS (GenHeader::w_u); the next sample's are requested a sample ahead (HBM latency)
    double upre[COOP_SLOTS];
    for (int sl = 0; sl < COOP_SLOTS; ++sl) {
        const int k = lig + LPI * sl;
        upre[sl] = (k < H.nu && A.T > 0) ? A.u[(i * A.T) * H.nu + k] : 0.0;
    }
    for (long long n = 0; n < A.T; ++n) {
        double *yn = A.y + (i * A.T + n) * H.ny;
        // inputs of this sample into LDS, the next sample's requested
        for (int sl = 0; sl < COOP_SLOTS; ++sl) {
            const int k = lig + LPI * sl;
            if (k < H.nu) W[c.O.u + k] = upre[sl];
        }
        wv::wave_fence();
        for (int sl = 0; sl < COOP_SLOTS; ++sl) {
            const int k = lig + LPI * sl;
            if (k < H.nu && n + 1 < A.T) upre[sl] = A.u[(i * A.T + n + 1) * H.nu + k];
        }
        const double *un = W + c.O.u;
        COOP_T(c, CT_REST);
        const bool alive = !dead;
        long long its_sample = 0;
        if (has_sub) {
            // p = dq x + eq u  (src/ACME.jl:678-683; a first sub-problem has no fqprev term)
            for (int r = lig; r < s.np; r += LPI) {
                double acc;
                if (c.ell) {
                    acc = coop_ell_dot(c.M, s.e_dq, r, W + c.O.x, 0.0);
                    acc = coop_ell_dot(c.M, s.e_eq, r, un, acc);
                } else {
                    acc = coop_dot(c.M + s.o_dq + r, s.np, W + c.O.x, H.nx, 0.0);
                    acc = coop_dot(c.M + s.o_eq + r, s.np, un, H.nu, acc);
                }
                if (alive) W[c.O.p + r] = acc;
            }
            wv::wave_fence();
            COOP_T(c, CT_PRE);
            int its;
            const bool conv = coop_homotopy_solve<NC>(c, s, f, alive, its);
            its_sample = alive ? its : 0;
            const bool failed = alive && !conv;
            if (wv::ballot(failed) != 0ull) {            // the policy of step! (src/ACME.jl:688-694)
                bool nf = false;
                for (int r = lig; r < s.nn; r += LPI) nf = nf || !(W[c.O.zz + r] * 0.0 == 0.0);
                const bool zfinite = !coop_any(c, nf);
                if (failed && lig == 0 && wr) {
                    if (zfinite) {
                        rep[RW_NWARN] += 1;
                        if (rep[RW_FIRST_NONCONV] < 0) rep[RW_FIRST_NONCONV] = A.sample_base + n;
                    } else if (rep[RW_FIRST_NONFINITE] < 0) {
                        rep[RW_FIRST_NONFINITE] = A.sample_base + n;
                    }
                }
                dead = dead || (failed && !zfinite);
            }
            for (int r = lig; r < s.nn; r += LPI)
                if (alive) W[c.O.z + s.zoff + r] = W[c.O.zz + r];
            wv::wave_fence();
        }
        it_total += its_sample;
        if (its_sample > it_max) it_max = its_sample;
        const bool live = !dead;
        COOP_T(c, CT_REST);
        // y = y0 + dy x + ey u + fy z (old x, :699-706);  x = x0 + a x + b u + c z (:708-714)
        // (ONE pass over the nx + ny rows -- a lane takes a state row or an output row, the vectors are the same: the two
        // loops cost two sets of latencies for mostly idle lanes)
        for (int rho = lig; rho < H.nx + H.ny; rho += LPI) {
            const bool isx = rho < H.nx;
            const int r = isx ? rho : rho - H.nx, ldm = isx ? H.nx : H.ny;
            double acc;
            if (c.ell) {
                acc = coop_ell_dot(c.M, H.e_ax, rho, W + c.O.x, c.M[H.o_xy0 + rho]);
                acc = coop_ell_dot(c.M, H.e_bu, rho, un, acc);
                acc = coop_ell_dot(c.M, H.e_cz, rho, W + c.O.z, acc);
            } else {
                acc = coop_dot(c.M + (isx ? H.o_a : H.o_dy) + r, ldm, W + c.O.x, H.nx, c.M[(isx ? H.o_x0 : H.o_y0) + r]);
                acc = coop_dot(c.M + (isx ? H.o_b : H.o_ey) + r, ldm, un, H.nu, acc);
                acc = coop_dot(c.M + (isx ? H.o_c : H.o_fy) + r, ldm, W + c.O.z, H.nnt, acc);
            }
            if (isx) {
                if (live) W[c.O.xn + r] = acc;
            } else {
                if (wr) yn[r] = live ? acc : (double)NAN;
            }
        }
        wv::wave_fence();
        for (int r = lig; r < H.nx; r += LPI)
            if (live) W[c.O.x + r] = W[c.O.xn + r];
        wv::wave_fence();
        COOP_T(c, CT_XY);
    }
#ifdef ACME_COOP_TIMING
    if (lig == 0 && wr && A.T >= CT_N && H.ny > 0)
        for (int k = 0; k < CT_N; ++k) A.y[(i * A.T + k) * H.ny] = (double)tmr.t[k];
#endif
    if (wr)
        for (int k = lig; k < H.nx; k += LPI) st[k] = W[c.O.x + k];
    if (has_sub && wr) {
        for (int j = lig; j < s.np; j += LPI) st[H.nx + s.poff + j] = W[c.O.lp + j];
        for (int r = lig; r < s.nn; r += LPI) st[H.nx + H.npt + s.zoff + r] = W[c.O.lz + r];
        if (caching)
            for (int k = lig; k < s.np * CACHE + 2; k += LPI) cache_g[k] = Cp[k];
        if constexpr (NC > 0)          // (a run split over several launches repeats the one-launch arithmetic)
            for (int sl = 0; sl < 2; ++sl)
                if (lig + LPI * sl < s.nn) A.coop_order[i * COOP_MAX_N + lig + LPI * sl] = f.rid[sl];
        if constexpr (NC < 0)
            for (int sl = 0; sl < coop_ns(NC); ++sl)
                if (lig + LPI * sl < s.nn) A.coop_order[i * COOP_MAX_N + lig + LPI * sl] = f.rid4[sl];
    }
    if (lig == 0 && wr) {
        rep[RW_ITERS_TOTAL] += it_total;
        if (it_max > rep[RW_ITERS_MAX]) rep[RW_ITERS_MAX] = it_max;
    }
}
#endif  // ACME_DEV

}  // namespace acme
