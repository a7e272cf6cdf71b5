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
// What makes it fast: an LDS instruction serves 4 instances x 16 rows; the O(nn^3) elimination's inner loop is three LDS
// operations per multiply-add with nothing but LDS latency behind it; 12 instances of 20 unknowns are resident per
// compute unit (12.6 KB of LDS each).
//
// Control flow is WAVE-UNIFORM throughout (the four instances of a wave iterate together, finished ones ride along with
// their writes predicated off), as in the tuned kernels: data-dependent trip counts are ballots.  Row interchanges of the
// partially pivoted LU are real (the rows sit in LDS, not in lanes); the sequence of interchanges is kept as the composed
// gather src[] (x_permuted[i] = x[src[i]]), which is what solve! applies first (src/solvers.jl:103-109).
//
// A wave never talks to another wave: blocks may be one wave (the GPU launch) or four (the CPU emulator's block).
#pragma once
#include "acme_generic.h"

namespace acme {

constexpr int COOP_MAX_N = 64;      // unknowns / parameters of the sub-problem (4 rows per lane)
constexpr int COOP_SLOTS = COOP_MAX_N / GROUP;

#ifdef ACME_DEV
// -DACME_COOP_TIMING (tools/coop_timing_probe.py): shader-clock cycles per code region, per wave, written over y's first samples
#ifdef ACME_COOP_TIMING
enum { CT_SETP, CT_EXTRAP, CT_EVAL, CT_LU, CT_SOLVE, CT_ACCEPT, CT_LOOKUP, CT_XY, CT_PRE, CT_REST, CT_N };
struct CoopTimer { long long t[CT_N]; long long mark; };
#define COOP_T(c, b) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = (long long)__builtin_readcyclecounter(); \
                          (c).tm->t[b] += t_ - (c).tm->mark; (c).tm->mark = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define COOP_T(c, b) do { } while (0)
#endif
struct CoopCtx {
    const GArgs &A;
    const GenHeader &H;
    const double *M;         // this instance's image: the block's copy in LDS (a shared image) or HBM (private images)
    double *W;               // this instance's workspace in LDS: the generic kernel's offsets (GenHeader::w_*)
    double *Cp;              // ... and its solution cache's stored p's and counters: cp[np][CACHE] | count, head (LDS)
    const double *tk;        // the block's copy of the row tables in LDS: k[8] of every row ([blk][8][16]) ...
    const int *ti;           // ... and the rows' ints ([blk][ROWI][16])
    int lig, grp;
    long long i;
    bool valid;
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
ACME_HD inline int coop_inst_doubles(const GenHeader &H) { return ((H.ws_total + 1) & ~1) + coop_cache_doubles(H); }
ACME_HD inline int coop_shared_doubles(const GenHeader &H, bool shared_image) {
    return (shared_image ? ((H.image_total + 1) & ~1) : 0) + coop_table_doubles(H);
}
ACME_DEV bool coop_any(const CoopCtx &c, bool x) { return ((wv::ballot(x) >> (c.grp * GROUP)) & 0xFFFFull) != 0ull; }

// Inner loops in BATCHES: a wave has nothing but its own instruction stream to hide a load's latency behind (LDS ~100
// cycles, the model image in L2 several hundred), and the compiler may not move a load across the store of the iteration
// before it (same array, run-time indices).  So every inner loop first requests COOP_B operands of each kind, then does
// its arithmetic in the reference's order, then stores: one latency per batch instead of one per multiply-add.
constexpr int COOP_B = 8;
// acc + sum_{j < n} a[j * sa] * b[j], accumulated in the order j = 0, 1, ... (fma chain, as the generic kernel's loops)
ACME_DEV double coop_dot(const double *a, int sa, const double *b, int n, double acc) {
    for (int j = 0; j < n; j += COOP_B) {
        double av[COOP_B], bv[COOP_B];
        for (int u = 0; u < COOP_B; ++u) {
            const int jj = j + u < n ? j + u : n - 1;      // (the tail re-reads the last operand: never out of range)
            av[u] = a[jj * sa];
            bv[u] = b[jj];
        }
        for (int u = 0; u < COOP_B; ++u)
            if (j + u < n) acc = fma(av[u], bv[u], acc);
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
    for (int r = c.lig; r < s.nq; r += GROUP) {
        c.W[c.H.w_pf + r] = coop_dot(c.M + s.o_pexp + r, s.nq, c.W + w_p, s.np, c.M[s.o_q0 + r]);
    }
    wv::wave_fence();
}

// evaluate!(nleq, z) (src/ACME.jl:178-188, src/circuit.jl:10-17): res, J (ROW-major at o_lu, row pitch GenHeader::ldf), the rows' Jq
// non-zeros.
// Returns (per lane) whether one of its residuals / Jacobian entries is not finite.
ACME_DEV bool coop_evaluate(const CoopCtx &c, const GenSub &s, int w_z, int o_lu) {
    const GenHeader &H = c.H;
    for (int r = c.lig; r < s.nq; r += GROUP) {
        c.W[H.w_q + r] = coop_dot(c.M + s.o_fq + r, s.nq, c.W + w_z, s.nn, c.W[H.w_pf + r]);
    }
    wv::wave_fence();
    bool bad = false;
    const wv::ExpTab etab = wv::load_exp_tab();          // (once per evaluate!, not twice per row)
    for (int r = c.lig; r < s.nn; r += GROUP) {
        RowDesc rd;
        int tc[4];
        coop_rowdesc(c, s.row0 + r, rd, tc);
        double e[4], tv[4], res;
        for (int t = 0; t < 4; ++t) e[t] = c.W[H.w_q + tc[t]];
        const bool expo = rd.kind == RK_DIODE || rd.kind == RK_BJT;
        const double exA = exp_junction(expo ? e[0] * rd.k[0] : 0.0, etab);
        const double exB = H.has_bjt ? exp_junction(rd.kind == RK_BJT ? e[1] * rd.k[1] : 0.0, etab) : 1.0;
        eval_row<true, 4>(rd, e, exA, exB, res, tv);
        c.W[H.w_res + r] = res;
        bad = bad || !(res * 0.0 == 0.0);
        for (int t = 0; t < 4; ++t) c.W[H.w_tv + 4 * r + t] = tv[t];
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
    for (int r = c.lig; r < s.nn; r += GROUP) {
        RowDesc rd;
        int tc[4];
        coop_rowdesc(c, s.row0 + r, rd, tc);
        double tv[4];
        for (int t = 0; t < 4; ++t) tv[t] = c.W[H.w_tv + 4 * r + t];
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
    const int ld = c.H.ldf;
    for (int i = c.lig; i < n; i += GROUP) W[o_src + i] = (double)i;
    wv::wave_fence();
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        double best = -1.0, bi = 1e9;
        for (int i = c.lig; i < n; i += GROUP)
            if (i >= k) {
                const double v = fabs(W[o_f + i * ld + k]);
                if (v > best) { best = v; bi = (double)i; }
            }
        const double m = wv::allmax16(best);
        // the reference starts from (amax = 0, kp = k) and takes the first strictly larger entry: the smallest index
        // holding the maximum; an all-zero (or all-NaN) column keeps kp = k
        double kpd = wv::allmin16((best == m && m > 0.0) ? bi : 1e9);
        const int kp = kpd < (double)n ? (int)kpd : k;
        const double piv = W[o_f + kp * ld + k];
        ok = ok && piv != 0.0;
        wv::wave_fence();
        if (kp != k) {
            for (int j = c.lig; j < n; j += GROUP) {
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
        for (int i = c.lig; i < n; i += GROUP)
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
    const int ld = c.H.ldf;
    const int ns = (n + GROUP - 1) / GROUP;          // slots in use (uniform): the others' code is skipped, not predicated
    double xs[COOP_SLOTS];
    sfor<0, COOP_SLOTS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        xs[sl] = 0.0;
        if (sl < ns) {
            const int i = c.lig + GROUP * sl;
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
                    const int i = c.lig + GROUP * sl;
                    f[sl] = (sl < ns && i > j && i < n) ? W[o_f + i * ld + j] : 0.0;
                });
                const double xj = wv::shfl16(xs[sj], jj);
                sfor<sj, COOP_SLOTS>([&](auto sc) ACME_LAMBDA {
                    constexpr int sl = decltype(sc)::value;
                    const int i = c.lig + GROUP * sl;
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
                    const int i = c.lig + GROUP * sl;
                    f[sl] = i < j ? W[o_f + i * ld + j] : 0.0;
                });
                const double xj = W[o_f + j * ld + j] * wv::shfl16(xs[sj], jj);
                sfor<0, sj + 1>([&](auto sc) ACME_LAMBDA {
                    constexpr int sl = decltype(sc)::value;
                    const int i = c.lig + GROUP * sl;
                    xs[sl] = i == j ? xj : (i < j ? xs[sl] - f[sl] * xj : xs[sl]);
                });
            }
    });
    wv::wave_fence();
    sfor<0, COOP_SLOTS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int i = c.lig + GROUP * sl;
        if (sl < ns && i < n) W[w_x + i] = xs[sl];
    });
    wv::wave_fence();
}

// ---- the factor matrix in REGISTERS (17 ... 32 unknowns: kernels instantiated per NC = the unknowns rounded up to 4) ----
// The LDS version above walks the elimination through LDS: per step a pivot search, an interchange and an update, each a
// round trip of LDS latency that a lone wave has nothing to hide behind (96 000 of 187 000 cycles per sample at 20 unknowns).
// Here a lane reads its (up to two) rows ONCE (ds_read_b128), keeps them in registers through all n steps -- column indices
// are compile-time constants, the step loop is written out -- and writes the factors once.  A row never moves: the
// reference's full-row interchange becomes the row's POSITION label (pos: where the row would sit after the interchanges so
// far), which is what decides a tie in the pivot search (first strict maximum = the smallest position) and where the row is
// written at the end; the step's pivot row travels through LDS row k of the result (its final place), one write by its
// holder and one broadcast read by everyone.  Arithmetic per entry: unchanged (l = a_ik * (1 / a_kk), a_ij -= l a_kj in
// the order k = 0, 1, ...), so the factors, the gather and the zero-pivot verdict are coop_lu's bit for bit.
constexpr int COOP_REG_SLOTS = 2;
template <int NC> ACME_DEV bool coop_lu_reg(const CoopCtx &c, int n, int o_f, int o_src) {
    static_assert(NC % 4 == 0 && NC <= GROUP * COOP_REG_SLOTS, "columns in pairs, two rows per lane");
    constexpr int NS = COOP_REG_SLOTS;
    double *W = c.W;
    const int ld = c.H.ldf;
    double *F = W + o_f;                      // 16-byte aligned rows (even offset, even pitch: acme_pack.h)
    double a[NS][NC];
    int pos[NS];
    bool real[NS];
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int i = c.lig + GROUP * sl;
        real[sl] = i < n;
        pos[sl] = i;
        sfor<0, NC / 2>([&](auto gc) ACME_LAMBDA {
            constexpr int g = decltype(gc)::value;
            wv::pair_t v{0.0, 0.0};
            if (real[sl]) v = wv::ld2(F + i * ld + 2 * g);
            a[sl][2 * g] = v.lo;
            a[sl][2 * g + 1] = v.hi;
        });
    });
    wv::wave_fence();
    bool ok = true;
    sfor<0, NC>([&](auto kc) ACME_LAMBDA {
        constexpr int k = decltype(kc)::value;
        if (k < n) {
            // the pivot: the largest |a_ik| among the rows at positions >= k, the smallest position among equals
            double best = -1.0;
            int bp = 1 << 30;
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                {
                    const double v = fabs(a[sl][k]);
                    const bool cand = real[sl] && pos[sl] >= k;
                    if (cand && (v > best || (v == best && pos[sl] < bp))) {
                        best = v;
                        bp = pos[sl];
                    }
                }
            });
            const double m = wv::allmax16(best);
            const double kpd = wv::allmin16((best == m && m > 0.0) ? (double)bp : 1e9);
            const int kp = kpd < (double)n ? (int)kpd : k;
            // its holder puts it where it belongs -- row k of the result (columns right of the diagonal are final) ...
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                const bool holds = real[sl] && pos[sl] == kp;
                if (holds)
                    sfor<k / 2, NC / 2>([&](auto gc) ACME_LAMBDA {
                        constexpr int g = decltype(gc)::value;
                        wv::st2(F + k * ld + 2 * g, a[sl][2 * g], a[sl][2 * g + 1]);
                    });
                // (the interchange: positions k and kp trade places)
                pos[sl] = holds ? k : (pos[sl] == k ? kp : pos[sl]);
            });
            wv::wave_fence();
            // ... and everyone reads it back
            double b[NC];
            sfor<k / 2, NC / 2>([&](auto gc) ACME_LAMBDA {
                constexpr int g = decltype(gc)::value;
                const wv::pair_t v = wv::ld2(F + k * ld + 2 * g);
                b[2 * g] = v.lo;
                b[2 * g + 1] = v.hi;
            });
            wv::wave_fence();
            const double piv = b[k];
            ok = ok && piv != 0.0;
            const double inv = 1.0 / piv;
            sfor<0, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                if (real[sl] && pos[sl] > k) {
                    const double l = a[sl][k] * inv;
                    a[sl][k] = l;
                    sfor<k + 1, NC>([&](auto jc) ACME_LAMBDA {
                        constexpr int j = decltype(jc)::value;
                        a[sl][j] -= l * b[j];
                    });
                }
                if (real[sl] && pos[sl] == k) a[sl][k] = inv;          // the reciprocal on the diagonal (src/solvers.jl:86)
            });
        }
    });
    wv::wave_fence();
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        if (real[sl]) {
            sfor<0, NC / 2>([&](auto gc) ACME_LAMBDA {
                constexpr int g = decltype(gc)::value;
                wv::st2(F + pos[sl] * ld + 2 * g, a[sl][2 * g], a[sl][2 * g + 1]);
            });
            W[o_src + pos[sl]] = (double)(c.lig + GROUP * sl);
        }
    });
    wv::wave_fence();
    return ok;
}

// solve! with a lane's rows of the factors in registers (read once, ds_read_b128) and x_j handed round by DPP broadcasts
// (row j of the final order sits in lane j mod 16: a compile-time lane): the 2 n sequential steps cost a broadcast and a
// multiply-add each, where coop_lu_solve's cost an LDS read and a ds_bpermute round trip.  Same arithmetic, same order.
template <int NC> ACME_DEV void coop_lu_solve_reg(const CoopCtx &c, int n, int o_f, int o_src, int w_x) {
    constexpr int NS = COOP_REG_SLOTS;
    double *W = c.W;
    const int ld = c.H.ldf;
    const double *F = W + o_f;
    double f[NS][NC], xs[NS];
    bool real[NS];
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        const int i = c.lig + GROUP * sl;
        real[sl] = i < n;
        xs[sl] = real[sl] ? W[w_x + (int)W[o_src + i]] : 0.0;
        sfor<0, NC / 2>([&](auto gc) ACME_LAMBDA {
            constexpr int g = decltype(gc)::value;
            wv::pair_t v{0.0, 0.0};
            if (real[sl]) v = wv::ld2(F + i * ld + 2 * g);
            f[sl][2 * g] = v.lo;
            f[sl][2 * g + 1] = v.hi;
        });
    });
    // forward: x_i -= F[i][j] x_j for i > j
    sfor<0, NC>([&](auto jc) ACME_LAMBDA {
        constexpr int j = decltype(jc)::value;
        if (j < n) {
            const double xj = wv::bcast16<j % GROUP>(xs[j / GROUP]);
            sfor<j / GROUP, NS>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                const int i = c.lig + GROUP * sl;
                const double t = xs[sl] - f[sl][j] * xj;
                xs[sl] = (real[sl] && i > j) ? t : xs[sl];
            });
        }
    });
    // backward: x_j *= 1 / F[j][j] (stored), x_i -= F[i][j] x_j for i < j
    sfor_down<NC>([&](auto jc) ACME_LAMBDA {
        constexpr int j = decltype(jc)::value;
        if (j < n) {
            const double xj = wv::bcast16<j % GROUP>(f[j / GROUP][j]) * wv::bcast16<j % GROUP>(xs[j / GROUP]);
            sfor<0, j / GROUP + 1>([&](auto sc) ACME_LAMBDA {
                constexpr int sl = decltype(sc)::value;
                const int i = c.lig + GROUP * sl;
                const double t = xs[sl] - f[sl][j] * xj;
                xs[sl] = i == j ? xj : (i < j ? t : xs[sl]);
            });
        }
    });
    wv::wave_fence();
    sfor<0, NS>([&](auto sc) ACME_LAMBDA {
        constexpr int sl = decltype(sc)::value;
        if (real[sl]) W[w_x + c.lig + GROUP * sl] = xs[sl];
    });
    wv::wave_fence();
}

// the factorisation / the solve of a kernel instantiated for NC columns (0: the LDS versions, any size)
template <int NC> ACME_DEV bool coop_factor(const CoopCtx &c, int n, int o_f, int o_src) {
    if constexpr (NC > 0) return coop_lu_reg<NC>(c, n, o_f, o_src);
    else return coop_lu(c, n, o_f, o_src);
}
template <int NC> ACME_DEV void coop_backsolve(const CoopCtx &c, int n, int o_f, int o_src, int w_x) {
    if constexpr (NC > 0) coop_lu_solve_reg<NC>(c, n, o_f, o_src, w_x);
    else coop_lu_solve(c, n, o_f, o_src, w_x);
}

// the solver of one instance's ONE sub-problem: where its current factors / its origin's factors sit (they trade places
// when an iterate is accepted: no copy of nn x nn doubles per sample)
struct CoopSolver {
    int o_lu, o_src;         // scratch of the running solve
    int o_llu, o_lsrc;       // the extrapolation origin's factors (last_linsolver, src/solvers.jl:191-196)
};
ACME_DEV void coop_accept_factors(CoopSolver &f, bool pred) {
    const int a = f.o_lu, b = f.o_src;
    f.o_lu = pred ? f.o_llu : f.o_lu;
    f.o_src = pred ? f.o_lsrc : f.o_src;
    f.o_llu = pred ? a : f.o_llu;
    f.o_lsrc = pred ? b : f.o_lsrc;
}

// set_extrapolation_origin(solver, p, z) (src/solvers.jl:183-196) at (w_lp, w_lz) for the instances with `pred`
template <int NC> ACME_DEV void coop_set_origin(const CoopCtx &c, const GenSub &s, CoopSolver &f, bool pred) {
    coop_set_p(c, s, s.w_lp);
    (void)coop_evaluate(c, s, s.w_lz, f.o_lu);
    (void)coop_factor<NC>(c, s.nn, f.o_lu, f.o_src);
    coop_calc_jp(c, s, s.w_ljp, pred);
    coop_accept_factors(f, pred);
}

// solve(::SimpleSolver, p) (src/solvers.jl:207-236) for the instances with `need`: p at w_p, z left in w_zz; returns
// hasconverged, needediterations in its
template <int NC> ACME_DEV bool coop_simple_solve(const CoopCtx &c, const GenSub &s, CoopSolver &f, int w_p, bool need, int &its) {
    const GenHeader &H = c.H;
    double *W = c.W;
    const int nn = s.nn, np = s.np;
    COOP_T(c, CT_REST);
    coop_set_p(c, s, w_p);
    COOP_T(c, CT_SETP);
    // z <- last_z - last_J \ (last_Jp (p - last_p))
    for (int r = c.lig; r < nn; r += GROUP) {
        double acc = 0.0;
        for (int j = 0; j < np; j += COOP_B) {
            double jv[COOP_B], pv[COOP_B], lv[COOP_B];
            for (int u = 0; u < COOP_B; ++u) {
                const int jj = j + u < np ? j + u : np - 1;
                jv[u] = W[s.w_ljp + jj * nn + r];
                pv[u] = W[w_p + jj];
                lv[u] = W[s.w_lp + jj];
            }
            for (int u = 0; u < COOP_B; ++u)
                if (j + u < np) acc = fma(jv[u], pv[u] - lv[u], acc);
        }
        W[H.w_tmp + r] = acc;
    }
    wv::wave_fence();
    coop_backsolve<NC>(c, nn, f.o_llu, f.o_lsrc, H.w_tmp);
    for (int r = c.lig; r < nn; r += GROUP)
        if (need) W[H.w_zz + r] = W[s.w_lz + r] - W[H.w_tmp + r];
    wv::wave_fence();
    COOP_T(c, CT_EXTRAP);
    bool act = need, conv = false;
    double reslast = 0.0;
    its = 0;
    while (wv::ballot(act) != 0ull) {
        its += act ? 1 : 0;
        const bool bad = coop_evaluate(c, s, H.w_zz, f.o_lu);
        COOP_T(c, CT_EVAL);
        const bool finite = !coop_any(c, bad);
        double rm = 0.0;
        for (int r = c.lig; r < nn; r += GROUP) {
            const double v = fabs(W[H.w_res + r]);
            if (v > rm) rm = v;
        }
        double resmax = wv::allmax16(rm);
        if (!finite) resmax = (double)NAN;
        const bool ok = coop_factor<NC>(c, nn, f.o_lu, f.o_src);
        COOP_T(c, CT_LU);
        const bool small = resmax < c.A.tol;
        const bool accept = act && finite && ok && small;
        const bool step = act && finite && ok && !small;
        reslast = act ? resmax : reslast;
        // the Newton step (for everyone; only the stepping instances keep it)
        for (int r = c.lig; r < nn; r += GROUP) W[H.w_dz + r] = W[H.w_res + r];
        wv::wave_fence();
        coop_backsolve<NC>(c, nn, f.o_lu, f.o_src, H.w_dz);
        for (int r = c.lig; r < nn; r += GROUP)
            if (step) W[H.w_zz + r] -= W[H.w_dz + r];
        COOP_T(c, CT_SOLVE);
        // an accepted iterate: its factors, Jp, p and z become the extrapolation origin
        if (wv::ballot(accept) != 0ull) {
            coop_calc_jp(c, s, s.w_ljp, accept);
            coop_accept_factors(f, accept);
            for (int j = c.lig; j < np; j += GROUP)
                if (accept) W[s.w_lp + j] = W[w_p + j];
            for (int r = c.lig; r < nn; r += GROUP)
                if (accept) W[s.w_lz + r] = W[H.w_zz + r];
        }
        wv::wave_fence();
        COOP_T(c, CT_ACCEPT);
        conv = conv || accept;
        act = step && its < c.A.maxiter;
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
    if (caching) {
        static_assert(CACHE == GROUP, "one stored solution per lane");
        double best = 0.0, d = 0.0;
        const int count = c.valid ? meta[0] : 0;
        for (int j = 0; j < np; ++j) {
            const double pj = W[w_p + j];
            const double dl = pj - W[s.w_lp + j];
            best = fma(dl, dl, best);
            const double t = (c.valid ? cp[j * CACHE + c.lig] : 0.0) - pj;
            d = fma(t, t, d);
        }
        d = c.lig < count ? d : (double)INFINITY;
        const double m = wv::allmin16(d);
        const unsigned long long bal = wv::ballot(d == m);
        const int idx = wv::ffs32((int)((bal >> (c.grp * GROUP)) & 0xFFFFull)) - 1;
        const bool hit = need && count > 0 && m < best;
        if (wv::ballot(hit) != 0ull) {
            const int e = hit ? idx : 0;
            for (int j = c.lig; j < np; j += GROUP)
                if (hit) W[s.w_lp + j] = cp[j * CACHE + e];
            for (int r = c.lig; r < nn; r += GROUP)
                if (hit) W[s.w_lz + r] = cz[e * nn + r];
            wv::wave_fence();
            coop_set_origin<NC>(c, s, f, hit);
        }
    }
    COOP_T(c, CT_LOOKUP);
    const bool conv = coop_simple_solve<NC>(c, s, f, w_p, need, its);
    if (caching) {
        const bool keep = need && conv && its > 5;
        if (wv::ballot(keep) != 0ull) {
            const int count = c.valid ? meta[0] : 0, head = c.valid ? meta[1] : 0;
            const int slot = count < CACHE ? count : head;
            wv::wave_fence();
            for (int j = c.lig; j < np; j += GROUP)
                if (keep) cp[j * CACHE + slot] = W[w_p + j];
            for (int r = c.lig; r < nn; r += GROUP)
                if (keep) cz[slot * nn + r] = W[c.H.w_zz + r];
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
    int w_src = H.w_p;
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
                for (int j = c.lig; j < s.np; j += GROUP)
                    if (need) W[H.w_sp + j] = W[s.w_lp + j];
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
        for (int j = c.lig; j < s.np; j += GROUP) {
            double pa = W[H.w_sp + j] * (1.0 - a);
            pa = pa + a * W[H.w_p + j];
            if (need) W[H.w_pa + j] = pa;
        }
        wv::wave_fence();
        w_src = H.w_pa;
    } while (true);
    return conv;
}

// run! for the instances of one wave (GArgs::mode == GEN_RUN).  lds: this wave's LDS (layout above); IMGL: the batch shares
// one model image, staged in LDS -- a dependent load from L2 costs a lone wave ~1 us, and evaluate! alone chains five of them.
template <bool IMGL, int NC> ACME_DEV void coop_main(const GArgs &A, double *lds, int wave_global, int lane) {
    const GenHeader &H = *A.H;
    const int lig = lane & (GROUP - 1), grp = lane >> 4;
    const int gpw = A.coop_gpw;
    lds = static_cast<double *>(__builtin_assume_aligned(lds, 16));
    // ---- the wave's shared part: model image (if shared) and row tables, loaded by all 64 lanes ----
    double *img = lds;
    double *tk = lds + (IMGL ? ((H.image_total + 1) & ~1) : 0);
    const int blocks = (H.nnt + GROUP - 1) / GROUP;
    int *ti = reinterpret_cast<int *>(tk + blocks * 8 * GROUP);
    if constexpr (IMGL)
        for (int k = lane; k < H.image_total; k += 64) img[k] = A.image[k];
    for (int k = lane; k < blocks * 8 * GROUP; k += 64) {
        const int blk = k / (8 * GROUP), rest = k % (8 * GROUP);
        tk[k] = A.rowc[(long long)blk * ROWC * GROUP + rest];       // constants 0 .. 7 of the block's 16 rows
    }
    for (int k = lane; k < blocks * ROWI * GROUP; k += 64) ti[k] = A.rowi[k];
    wv::wave_fence();
    const long long slot = (long long)wave_global * gpw + grp;
    const bool valid = grp < gpw && slot < A.n_inst;
    if (!valid) return;      // (a row of 16 lanes without an instance leaves: nothing below crosses the rows of a wave)
    const long long i = slot;
    double *W = lds + coop_shared_doubles(H, IMGL) + (long long)grp * coop_inst_doubles(H);
    double *Cp = W + ((H.ws_total + 1) & ~1);
    CoopCtx c{A, H, IMGL ? img : A.image + i * A.image_stride, W, Cp, tk, ti, lig, grp, i, valid};
#ifdef ACME_COOP_TIMING
    CoopTimer tmr{};
    tmr.mark = (long long)__builtin_readcyclecounter();
    c.tm = &tmr;
#endif
    double *st = A.state + i * H.state_total;
    long long *rep = A.report + i * RW_WORDS;
    const bool has_sub = H.nsub > 0;
    const GenSub &s = H.sub[0];
    CoopSolver f{H.w_lu, H.w_piv, has_sub ? s.w_llu : 0, has_sub ? s.w_lpiv : 0};
    const bool caching = has_sub && A.solver == SOLVER_CACHING_HOMOTOPY;
    double *cache_g = A.cache + i * H.cache_total + (has_sub ? s.c_off : 0);
    for (int k = lig; k < H.ws_total; k += GROUP) W[k] = 0.0;
    wv::wave_fence();
    for (int k = lig; k < H.nx; k += GROUP) W[H.w_x + k] = st[k];
    if (has_sub) {
        for (int j = lig; j < s.np; j += GROUP) W[s.w_lp + j] = st[H.nx + s.poff + j];
        for (int r = lig; r < s.nn; r += GROUP) W[s.w_lz + r] = st[H.nx + H.npt + s.zoff + r];
        if (caching)       // the stored p's and the two counters live in LDS for the launch
            for (int k = lig; k < s.np * CACHE + 2; k += GROUP) Cp[k] = cache_g[k];
    }
    wv::wave_fence();
    if (has_sub) coop_set_origin<NC>(c, s, f, true);
    bool dead = rep[RW_FIRST_NONFINITE] >= 0;
    long long it_total = 0, it_max = 0;
    // this sample's inputs sit in LDS (GenHeader::w_u); the next sample's are requested a sample ahead (HBM latency)
    double upre[COOP_SLOTS];
    for (int sl = 0; sl < COOP_SLOTS; ++sl) {
        const int k = lig + GROUP * sl;
        upre[sl] = (k < H.nu && A.T > 0) ? A.u[(i * A.T) * H.nu + k] : 0.0;
    }
    for (long long n = 0; n < A.T; ++n) {
        double *yn = A.y + (i * A.T + n) * H.ny;
        // inputs of this sample into LDS, the next sample's requested
        for (int sl = 0; sl < COOP_SLOTS; ++sl) {
            const int k = lig + GROUP * sl;
            if (k < H.nu) W[H.w_u + k] = upre[sl];
        }
        wv::wave_fence();
        for (int sl = 0; sl < COOP_SLOTS; ++sl) {
            const int k = lig + GROUP * sl;
            if (k < H.nu && n + 1 < A.T) upre[sl] = A.u[(i * A.T + n + 1) * H.nu + k];
        }
        const double *un = W + H.w_u;
        COOP_T(c, CT_REST);
        const bool alive = !dead;
        long long its_sample = 0;
        if (has_sub) {
            // p = dq x + eq u  (src/ACME.jl:678-683; a first sub-problem has no fqprev term)
            for (int r = lig; r < s.np; r += GROUP) {
                double acc = coop_dot(c.M + s.o_dq + r, s.np, W + H.w_x, H.nx, 0.0);
                acc = coop_dot(c.M + s.o_eq + r, s.np, un, H.nu, acc);
                if (alive) W[H.w_p + r] = acc;
            }
            wv::wave_fence();
            COOP_T(c, CT_PRE);
            int its;
            const bool conv = coop_homotopy_solve<NC>(c, s, f, alive, its);
            its_sample = alive ? its : 0;
            const bool failed = alive && !conv;
            if (wv::ballot(failed) != 0ull) {            // the policy of step! (src/ACME.jl:688-694)
                bool nf = false;
                for (int r = lig; r < s.nn; r += GROUP) nf = nf || !(W[H.w_zz + r] * 0.0 == 0.0);
                const bool zfinite = !coop_any(c, nf);
                if (failed && lig == 0) {
                    if (zfinite) {
                        rep[RW_NWARN] += 1;
                        if (rep[RW_FIRST_NONCONV] < 0) rep[RW_FIRST_NONCONV] = A.sample_base + n;
                    } else if (rep[RW_FIRST_NONFINITE] < 0) {
                        rep[RW_FIRST_NONFINITE] = A.sample_base + n;
                    }
                }
                dead = dead || (failed && !zfinite);
            }
            for (int r = lig; r < s.nn; r += GROUP)
                if (alive) W[H.w_z + s.zoff + r] = W[H.w_zz + r];
            wv::wave_fence();
        }
        it_total += its_sample;
        if (its_sample > it_max) it_max = its_sample;
        const bool live = !dead;
        COOP_T(c, CT_REST);
        // y = y0 + dy x + ey u + fy z (old x, :699-706);  x = x0 + a x + b u + c z (:708-714)
        for (int r = lig; r < H.ny; r += GROUP) {
            double acc = coop_dot(c.M + H.o_dy + r, H.ny, W + H.w_x, H.nx, c.M[H.o_y0 + r]);
            acc = coop_dot(c.M + H.o_ey + r, H.ny, un, H.nu, acc);
            acc = coop_dot(c.M + H.o_fy + r, H.ny, W + H.w_z, H.nnt, acc);
            yn[r] = live ? acc : (double)NAN;
        }
        for (int r = lig; r < H.nx; r += GROUP) {
            double acc = coop_dot(c.M + H.o_a + r, H.nx, W + H.w_x, H.nx, c.M[H.o_x0 + r]);
            acc = coop_dot(c.M + H.o_b + r, H.nx, un, H.nu, acc);
            acc = coop_dot(c.M + H.o_c + r, H.nx, W + H.w_z, H.nnt, acc);
            if (live) W[H.w_xn + r] = acc;
        }
        wv::wave_fence();
        for (int r = lig; r < H.nx; r += GROUP)
            if (live) W[H.w_x + r] = W[H.w_xn + r];
        wv::wave_fence();
        COOP_T(c, CT_XY);
    }
#ifdef ACME_COOP_TIMING
    if (lig == 0 && A.T >= CT_N && H.ny > 0)
        for (int k = 0; k < CT_N; ++k) A.y[(i * A.T + k) * H.ny] = (double)tmr.t[k];
#endif
    for (int k = lig; k < H.nx; k += GROUP) st[k] = W[H.w_x + k];
    if (has_sub) {
        for (int j = lig; j < s.np; j += GROUP) st[H.nx + s.poff + j] = W[s.w_lp + j];
        for (int r = lig; r < s.nn; r += GROUP) st[H.nx + H.npt + s.zoff + r] = W[s.w_lz + r];
        if (caching)
            for (int k = lig; k < s.np * CACHE + 2; k += GROUP) cache_g[k] = Cp[k];
    }
    if (lig == 0) {
        rep[RW_ITERS_TOTAL] += it_total;
        if (it_max > rep[RW_ITERS_MAX]) rep[RW_ITERS_MAX] = it_max;
    }
}
#endif  // ACME_DEV

}  // namespace acme
