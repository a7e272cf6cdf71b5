// acme_generic.h -- the kernel that never refuses a model: one LANE per circuit instance, every dimension a
// run-time number, every working array in HBM (interleaved over the instances so that the lanes of a wave touch
// consecutive addresses).  It runs what the tuned 16-lane kernels (acme_kernel.h) are not built for -- more than 16
// unknowns or 16 parameters in a sub-problem, more than 32 states, more than 4 nonlinear sub-problems, ... -- the way
// the reference runs ANY model it has derived (src/ACME.jl:150,349-378; its LU is meant for "sizes up to about
// 60 x 60", src/solvers.jl:53-54).  Slower by orders of magnitude than the tuned kernels, and the same solver
// stack operation for operation:
//   step!           src/ACME.jl:666-715      closures       src/ACME.jl:176-194,236-252
//   LinearSolver    src/solvers.jl:46-132    SimpleSolver   src/solvers.jl:151-236
//   HomotopySolver  src/solvers.jl:247-302   CachingSolver  src/solvers.jl:319-396 (bounded store: the last CACHE
//                                                            solutions, as in the tuned kernels)
// Element rows: eval_row<true, 4> of acme_kernel.h (every element kind).  State, reports, solution caches and the
// C ABI are the tuned kernels'; (x, last_p, last_z) persist between launches and the factors at the origin are
// recomputed from them at launch start (set_extrapolation_origin, src/solvers.jl:183-196), so a run split over
// several launches repeats the one-launch arithmetic.
#pragma once
#include "acme_common.h"

namespace acme {

constexpr int GEN_MAX_SUB = 64;

// The non-zeros of a matrix, row by row, for the mid-size kernel (acme_coop.h): entry e of row r at [e * rows + r] of the
// values and of the column numbers (kept as doubles), k entries per row -- the fullest row's count; rows with fewer end in
// (0.0, column 0).  A circuit's matrices are sparse (the 34-unknown clipper chain: 1.75 non-zeros per row of fq), a dense
// matrix-vector product reads mostly zeros -- from L2, when the image does not fit the LDS.  Taken in ascending column
// order the products that remain are the dense loop's, bit for bit: x + 0 * y = x.
struct GenEll { int o_val, o_col, k, rows; };
// one nonlinear sub-problem: dimensions, where its matrices sit in the model image (column-major, as the caller
// handed them over), its rows in the row tables, its slices of the per-instance arrays
struct GenSub {
    int nn, nq, np;
    int zoff;                // its unknowns inside the model's z (columns of c / fy / fqprev)
    int poff;                // its parameters inside the concatenated p of the state
    int o_pexp, o_dq, o_eq, o_fqprev, o_fq, o_q0;      // image offsets (doubles)
    int row0;                // first of its rows in the row tables (global row R: block R / 16, lane R % 16)
    int w_lp, w_lz, w_ljp, w_llu, w_lpiv;              // workspace: the extrapolation origin (p, z, Jp, LU, pivots)
    int c_off;               // its solution cache inside an instance's cache block
    // sparse forms (GenHeader::ell; only built for a model with ONE sub-problem):
    GenEll e_fq, e_pexp, e_dq, e_eq;
    int o_q0s;               // q0 once more, inside the sparse part of the image
    // J = Jq fq row by row: residual row r's columns with anything in them -- entry e: column jcol[e * nn + r] (padding: the
    // slack column nn + 1) and, for each of the row's four Jq terms t, fq[tc[t]][column] at jcoef[(e * 4 + t) * nn + r]
    int o_jcol, o_jcoef, kj;
    // Jp = Jq pexp likewise (padding: column np, a column of zeros the mid-size kernel keeps behind its Jp)
    int o_pcol, o_pcoef, kp;
};
struct GenHeader {
    int nx, nu, ny, nsub, nnt, npt;
    int o_a, o_b, o_c, o_x0, o_dy, o_ey, o_fy, o_y0;
    int image_total;
    int ell;                 // the sparse forms below and in sub[0] exist and hold every non-zero of this model (a batch: of every
                             // instance's model) -- the mid-size kernel reads them instead of the dense matrices
    int o_ell;               // where the sparse part of the image begins (an even offset; image_total if there is none)
    int o_jtab;              // ... and, inside it, where the rows of J and Jp begin (behind the matrices' own sparse forms)
    int reg_sparse;          // the mid-size kernel's REGISTER instantiations read the matrices' sparse forms for their matrix-vector
                             // products (set by the host where staging them beside the dense matrices costs no resident instance)
    int jp_sparse;           // the mid-size kernel on a matrix in LDS keeps Jp as the kp entries per row of its sparse form, not as
                             // np columns (set by the host for a batch that shares ONE model image: its sparse forms cannot change)
    GenEll e_ax, e_bu, e_cz; // [a; dy], [b; ey], [c; fy]: the nx state rows followed by the ny output rows
    int o_xy0;               // x0 followed by y0
    int nnmax, nqmax, npmax;
    int ldf;                 // leading dimension of the nnmax x nnmax factor matrices in the workspace (>= nnmax; the lane-per-instance
                             // kernel stores them column-major and packed, the cooperative one row-major with this row pitch)
    int has_bjt;
    // workspace (doubles per instance): x | xnew | zall | per sub-problem origins | scratch of one solve
    int w_x, w_xn, w_z, w_p, w_pa, w_sp, w_zz, w_res, w_dz, w_lu, w_piv, w_jp, w_q, w_pf, w_tv, w_tmp, w_u;
    int ws_total;
    int state_total;         // doubles per instance in the state array: x | last_p of every sub | last_z of every sub
    int cache_total;         // doubles per instance of solution caches: per sub cp[np][CACHE] | count, head | cz[CACHE][nn]
    GenSub sub[GEN_MAX_SUB];
};

enum { GEN_RUN = 0, GEN_SOLVE = 1, GEN_JAC = 2 };

struct GArgs {
    const GenHeader *H;
    const double *image;     // model image(s)
    long long image_stride;  // 0: shared; else doubles between per-instance images
    const double *rowc;      // row constants, blocks of 16 rows: [(R / 16) * ROWC + c] * 16 + R % 16
    const int *rowi;         // row ints, likewise with ROWI
    long long table_stride, tablei_stride;      // per-instance element tables: doubles / ints between them (0: shared)
    const double *u;         // [n_inst][T][nu]
    double *y;               // [n_inst][T][ny]
    double *state;           // [n_inst][state_total]
    long long *report;       // [n_inst][RW_WORDS]
    double *cache;           // [n_inst][cache_total]
    double *ws;              // [ws_total][n_inst]  (interleaved)
    long long n_inst, T, sample_base;
    double tol;
    int maxiter, solver, mode;
    const double *p_in;      // GEN_SOLVE: [n_inst][np_sub]
    double *z_out;           //            [n_inst][nn_sub]
    int *conv_out, *iters_out;
    double *jac_out;         // GEN_JAC:   [n_inst][np_sub][nn_sub]
    int solve_sub;
    int coop_imgl;           // acme_coop.h: the (shared) model image is staged in LDS
    int coop_gpw;            // acme_coop.h: instances per wave (4, 2 or 1: what the LDS of a compute unit holds most of)
    int coop_wpb;            // acme_coop.h: waves per block (4, 2 or 1: they share one copy of the row tables / the image in LDS)
    int coop_nc;             // acme_coop.h: the kernel's instantiation: 20, 24, 28, 32 -- that many columns of the Jacobian in registers
                             // (17 ... 32 unknowns rounded up to four; threshold pivoting in a learnt row order); -1 ... -4: the same
                             // elimination on one matrix in LDS, that many rows per lane; 0: the any-size kernel (the reference's LU
                             // literally, factors in LDS)
    int *coop_order;         // [n_inst][64]: threshold path, the row each position of an instance holds (kept between launches)
};

#ifdef ACME_DEV
// ---------------------------------------------------------------------------------------------------------------
// device code (the including translation unit provides ACME_DEV, namespace wv, and acme_kernel.h's eval_row)
// ---------------------------------------------------------------------------------------------------------------
struct GenCtx {
    const GArgs &A;
    const GenHeader &H;
    const double *M;         // this instance's image
    long long i, N;
    ACME_DEV double &W(int k) const { return A.ws[(long long)k * N + i]; }
};

// rows' descriptors out of the row tables
ACME_DEV void gen_rowdesc(const GArgs &A, int R, RowDesc &rd, int (&tc)[4], long long inst = 0) {
    const int blk = R / GROUP, ln = R % GROUP;
    const int *ri = A.rowi + inst * A.tablei_stride + (long long)blk * ROWI * GROUP + ln;
    rd.kind = ri[0 * GROUP];
    rd.erow = ri[1 * GROUP];
    rd.flags = ri[2 * GROUP];
    for (int t = 0; t < 4; ++t) tc[t] = ri[(3 + t) * GROUP];
    rd.rc = A.rowc + inst * A.table_stride + (long long)blk * ROWC * GROUP + ln;
    for (int c = 0; c < 8; ++c) rd.k[c] = rd.rc[c * GROUP];
}

// pfull <- q0 + pexp p  (set_p closure, src/ACME.jl:237-243)
ACME_DEV void gen_set_p(const GenCtx &c, const GenSub &s, int w_p) {
    for (int r = 0; r < s.nq; ++r) {
        double acc = c.M[s.o_q0 + r];
        for (int j = 0; j < s.np; ++j) acc = fma(c.M[s.o_pexp + j * s.nq + r], c.W(w_p + j), acc);
        c.W(c.H.w_pf + r) = acc;
    }
}

// evaluate!(nleq, z): q = pfull + fq z; (res, Jq) = elements(q); J = Jq fq  (src/ACME.jl:178-188, src/circuit.jl:10-17).
// Leaves res, J (column-major nn x nn in w_lu) and the rows' Jq non-zeros (w_tv).
ACME_DEV void gen_evaluate(const GenCtx &c, const GenSub &s, int w_z) {
    const GenHeader &H = c.H;
    for (int r = 0; r < s.nq; ++r) {
        double acc = c.W(H.w_pf + r);
        for (int j = 0; j < s.nn; ++j) acc = fma(c.M[s.o_fq + j * s.nq + r], c.W(w_z + j), acc);
        c.W(H.w_q + r) = acc;
    }
    for (int r = 0; r < s.nn; ++r) {
        RowDesc rd;
        int tc[4];
        gen_rowdesc(c.A, s.row0 + r, rd, tc, c.i);
        double e[4], tv[4], res;
        for (int t = 0; t < 4; ++t) e[t] = c.W(H.w_q + tc[t]);
        const bool expo = rd.kind == RK_DIODE || rd.kind == RK_BJT;
        const double exA = exp_junction(expo ? e[0] * rd.k[0] : 0.0);
        const double exB = H.has_bjt ? exp_junction(rd.kind == RK_BJT ? e[1] * rd.k[1] : 0.0) : 1.0;
        eval_row<true, 4>(rd, e, exA, exB, res, tv);
        c.W(H.w_res + r) = res;
        for (int t = 0; t < 4; ++t) c.W(H.w_tv + 4 * r + t) = tv[t];
        for (int j = 0; j < s.nn; ++j) {
            // (a row's Jq non-zeros may name the same q row twice -- padding terms carry a zero derivative)
            double acc = 0.0;
            for (int t = 0; t < 4; ++t) acc = fma(tv[t], c.M[s.o_fq + j * s.nq + tc[t]], acc);
            c.W(H.w_lu + j * s.nn + r) = acc;
        }
    }
}

// calc_Jp closure (src/ACME.jl:246-251): Jp = Jq pexp, with the Jq of the latest evaluate!
ACME_DEV void gen_calc_jp(const GenCtx &c, const GenSub &s) {
    const GenHeader &H = c.H;
    for (int r = 0; r < s.nn; ++r) {
        RowDesc rd;
        int tc[4];
        gen_rowdesc(c.A, s.row0 + r, rd, tc, c.i);
        for (int j = 0; j < s.np; ++j) {
            double acc = 0.0;
            for (int t = 0; t < 4; ++t) acc = fma(c.W(H.w_tv + 4 * r + t), c.M[s.o_pexp + j * s.nq + tc[t]], acc);
            c.W(H.w_jp + j * s.nn + r) = acc;
        }
    }
}

// setlhs! (src/solvers.jl:46-96) in place on the n x n matrix at w_f: partial pivoting, first strict maximum,
// full-row interchange, reciprocal on the diagonal; false on an exactly zero pivot
ACME_DEV bool gen_lu(const GenCtx &c, int n, int w_f, int w_piv) {
    for (int k = 0; k < n; ++k) {
        int kp = k;
        double amax = 0.0;
        for (int i = k; i < n; ++i) {
            const double v = fabs(c.W(w_f + k * n + i));
            if (v > amax) { kp = i; amax = v; }
        }
        c.W(w_piv + k) = (double)kp;
        if (c.W(w_f + k * n + kp) == 0.0) return false;
        if (k != kp)
            for (int j = 0; j < n; ++j) {
                const double t = c.W(w_f + j * n + k);
                c.W(w_f + j * n + k) = c.W(w_f + j * n + kp);
                c.W(w_f + j * n + kp) = t;
            }
        const double inv = 1.0 / c.W(w_f + k * n + k);
        c.W(w_f + k * n + k) = inv;
        for (int i = k + 1; i < n; ++i) c.W(w_f + k * n + i) *= inv;
        for (int j = k + 1; j < n; ++j) {
            const double fkj = c.W(w_f + j * n + k);
            for (int i = k + 1; i < n; ++i) c.W(w_f + j * n + i) -= c.W(w_f + k * n + i) * fkj;
        }
    }
    return true;
}
// solve! (src/solvers.jl:98-132), x in place
ACME_DEV void gen_lu_solve(const GenCtx &c, int n, int w_f, int w_piv, int w_x) {
    for (int i = 0; i < n; ++i) {
        const int p = (int)c.W(w_piv + i);
        const double t = c.W(w_x + i);
        c.W(w_x + i) = c.W(w_x + p);
        c.W(w_x + p) = t;
    }
    for (int j = 0; j < n; ++j) {
        const double xj = c.W(w_x + j);
        for (int i = j + 1; i < n; ++i) c.W(w_x + i) -= c.W(w_f + j * n + i) * xj;
    }
    for (int j = n - 1; j >= 0; --j) {
        const double xj = c.W(w_f + j * n + j) * c.W(w_x + j);
        c.W(w_x + j) = xj;
        for (int i = 0; i < j; ++i) c.W(w_x + i) -= c.W(w_f + j * n + i) * xj;
    }
}

ACME_DEV void gen_copy(const GenCtx &c, int dst, int src, int n) {
    for (int k = 0; k < n; ++k) c.W(dst + k) = c.W(src + k);
}

// set_extrapolation_origin(solver, p, z) (src/solvers.jl:183-196): linearise at the origin held in (w_lp, w_lz)
ACME_DEV void gen_set_origin(const GenCtx &c, const GenSub &s) {
    const GenHeader &H = c.H;
    gen_set_p(c, s, s.w_lp);
    gen_evaluate(c, s, s.w_lz);
    (void)gen_lu(c, s.nn, H.w_lu, H.w_piv);
    gen_calc_jp(c, s);
    gen_copy(c, s.w_llu, H.w_lu, s.nn * s.nn);
    gen_copy(c, s.w_lpiv, H.w_piv, s.nn);
    gen_copy(c, s.w_ljp, H.w_jp, s.nn * s.np);
}

// solve(::SimpleSolver, p) (src/solvers.jl:207-236): p at w_p; z in w_zz; returns hasconverged, iterations in its
ACME_DEV bool gen_simple_solve(const GenCtx &c, const GenSub &s, int w_p, int &its) {
    const GenHeader &H = c.H;
    const int nn = s.nn, np = s.np;
    gen_set_p(c, s, w_p);
    // z <- last_z - last_J \ (last_Jp (p - last_p))
    for (int r = 0; r < nn; ++r) {
        double acc = 0.0;
        for (int j = 0; j < np; ++j) acc = fma(c.W(s.w_ljp + j * nn + r), c.W(w_p + j) - c.W(s.w_lp + j), acc);
        c.W(H.w_tmp + r) = acc;
    }
    gen_lu_solve(c, nn, s.w_llu, s.w_lpiv, H.w_tmp);
    for (int r = 0; r < nn; ++r) c.W(H.w_zz + r) = c.W(s.w_lz + r) - c.W(H.w_tmp + r);
    bool conv = false;
    double resmax = 0.0;
    for (its = 1; its <= c.A.maxiter; ++its) {
        gen_evaluate(c, s, H.w_zz);
        resmax = 0.0;
        bool finite = true;
        for (int r = 0; r < nn; ++r) {
            const double v = fabs(c.W(H.w_res + r));
            if (!(v * 0.0 == 0.0)) finite = false;
            if (v > resmax) resmax = v;
        }
        if (finite)
            for (int k = 0; k < nn * nn; ++k)
                if (!(c.W(H.w_lu + k) * 0.0 == 0.0)) { finite = false; break; }
        if (!finite) { resmax = (double)NAN; break; }
        if (!gen_lu(c, nn, H.w_lu, H.w_piv)) break;                 // J singular: hasconverged is the residual test
        if (resmax < c.A.tol) { conv = true; break; }
        gen_copy(c, H.w_dz, H.w_res, nn);
        gen_lu_solve(c, nn, H.w_lu, H.w_piv, H.w_dz);
        for (int r = 0; r < nn; ++r) c.W(H.w_zz + r) -= c.W(H.w_dz + r);
    }
    if (its > c.A.maxiter) its = c.A.maxiter;
    const bool has = resmax < c.A.tol;          // hasconverged (:203): false for a NaN residual
    if (conv) {                                   // the factors belong to the converged point: new origin
        gen_calc_jp(c, s);
        gen_copy(c, s.w_llu, H.w_lu, nn * nn);
        gen_copy(c, s.w_lpiv, H.w_piv, nn);
        gen_copy(c, s.w_ljp, H.w_jp, nn * np);
        gen_copy(c, s.w_lp, w_p, np);
        gen_copy(c, s.w_lz, H.w_zz, nn);
    }
    return has;
}

// solve(::CachingSolver, p) (src/solvers.jl:347-396) with the bounded store; cache block of this sub-problem in HBM
ACME_DEV bool gen_cached_solve(const GenCtx &c, const GenSub &s, int w_p, int &its) {
    const int nn = s.nn, np = s.np;
    double *cp = c.A.cache + c.i * c.H.cache_total + s.c_off;
    int *meta = reinterpret_cast<int *>(cp + np * CACHE);
    double *cz = cp + np * CACHE + 2;
    const bool caching = c.A.solver == SOLVER_CACHING_HOMOTOPY;
    if (caching) {
        double best = 0.0;
        for (int j = 0; j < np; ++j) { const double d = c.W(w_p + j) - c.W(s.w_lp + j); best = fma(d, d, best); }
        int idx = -1;
        const int count = meta[0];
        for (int e = 0; e < count; ++e) {
            double d = 0.0;
            for (int j = 0; j < np; ++j) { const double t = cp[j * CACHE + e] - c.W(w_p + j); d = fma(t, t, d); }
            if (d < best) { best = d; idx = e; }
        }
        if (idx >= 0) {
            for (int j = 0; j < np; ++j) c.W(s.w_lp + j) = cp[j * CACHE + idx];
            for (int r = 0; r < nn; ++r) c.W(s.w_lz + r) = cz[idx * nn + r];
            gen_set_origin(c, s);
        }
    }
    const bool conv = gen_simple_solve(c, s, w_p, its);
    if (caching && conv && its > 5) {
        const int count = meta[0], head = meta[1];
        const int slot = count < CACHE ? count : head;
        for (int j = 0; j < np; ++j) cp[j * CACHE + slot] = c.W(w_p + j);
        for (int r = 0; r < nn; ++r) cz[slot * nn + r] = c.W(c.H.w_zz + r);
        meta[0] = count < CACHE ? count + 1 : count;
        meta[1] = count < CACHE ? head : (head + 1) & (CACHE - 1);
    }
    return conv;
}

// solve(::HomotopySolver, p) (src/solvers.jl:268-296); p at w_p
ACME_DEV bool gen_homotopy_solve(const GenCtx &c, const GenSub &s, int &its_total) {
    const GenHeader &H = c.H;
    int its;
    bool conv = gen_cached_solve(c, s, H.w_p, its);
    its_total = its;
    if (!conv && c.A.solver != SOLVER_SIMPLE) {
        double a = 0.5, best = 0.0;
        gen_copy(c, H.w_sp, s.w_lp, s.np);
        while (best < 1.0) {
            for (int j = 0; j < s.np; ++j) {
                double pa = c.W(H.w_sp + j) * (1.0 - a);
                pa = pa + a * c.W(H.w_p + j);
                c.W(H.w_pa + j) = pa;
            }
            conv = gen_cached_solve(c, s, H.w_pa, its);
            its_total += its;
            if (conv) {
                best = a;
                a = 1.0;
            } else {
                const double na = (a + best) / 2.0;
                if (!(best < na && na < a)) break;
                a = na;
            }
        }
    }
    return conv;
}

ACME_DEV void gen_main(const GArgs &A, long long i) {
    const GenHeader &H = *A.H;
    GenCtx c{A, H, A.image + (A.image_stride ? i * A.image_stride : 0), i, A.n_inst};
    double *st = A.state + i * H.state_total;
    long long *rep = A.report + i * RW_WORDS;
    for (int k = 0; k < H.nx; ++k) c.W(H.w_x + k) = st[k];
    for (int s_ = 0; s_ < H.nsub; ++s_) {
        const GenSub &s = H.sub[s_];
        for (int j = 0; j < s.np; ++j) c.W(s.w_lp + j) = st[H.nx + s.poff + j];
        for (int r = 0; r < s.nn; ++r) c.W(s.w_lz + r) = st[H.nx + H.npt + s.zoff + r];
        if (A.mode != GEN_RUN && s_ != A.solve_sub) continue;
        gen_set_origin(c, s);
    }
    if (A.mode == GEN_JAC) {            // get_extrapolation_jacobian = -(J \ Jp) at the origin (src/solvers.jl:198-201)
        const GenSub &s = H.sub[A.solve_sub];
        bool ok = true;
        for (int k = 0; k < s.nn; ++k) ok = ok && c.W(s.w_llu + k * s.nn + k) * 0.0 == 0.0;
        for (int j = 0; j < s.np; ++j) {
            gen_copy(c, H.w_tmp, s.w_ljp + j * s.nn, s.nn);
            gen_lu_solve(c, s.nn, s.w_llu, s.w_lpiv, H.w_tmp);
            for (int r = 0; r < s.nn; ++r) {
                const double v = -c.W(H.w_tmp + r);
                A.jac_out[(i * s.np + j) * s.nn + r] = (ok && v * 0.0 == 0.0) ? v : (double)NAN;
            }
        }
        return;
    }
    if (A.mode == GEN_SOLVE) {          // the solver-plugin contract: one solve(solver, p) on sub-problem solve_sub
        const GenSub &s = H.sub[A.solve_sub];
        for (int j = 0; j < s.np; ++j) c.W(H.w_p + j) = A.p_in[i * s.np + j];
        int its;
        const bool conv = gen_homotopy_solve(c, s, its);
        for (int r = 0; r < s.nn; ++r) A.z_out[i * s.nn + r] = c.W(H.w_zz + r);
        A.conv_out[i] = conv ? 1 : 0;
        A.iters_out[i] = its;
        for (int j = 0; j < s.np; ++j) st[H.nx + s.poff + j] = c.W(s.w_lp + j);
        for (int r = 0; r < s.nn; ++r) st[H.nx + H.npt + s.zoff + r] = c.W(s.w_lz + r);
        return;
    }
    bool dead = rep[RW_FIRST_NONFINITE] >= 0;
    long long it_total = 0, it_max = 0;
    for (long long n = 0; n < A.T; ++n) {
        const double *un = A.u + (i * A.T + n) * H.nu;
        double *yn = A.y + (i * A.T + n) * H.ny;
        if (dead) {
            for (int k = 0; k < H.ny; ++k) yn[k] = (double)NAN;
            continue;
        }
        for (int k = 0; k < H.nnt; ++k) c.W(H.w_z + k) = 0.0;
        long long its_sample = 0;
        for (int s_ = 0; s_ < H.nsub && !dead; ++s_) {
            const GenSub &s = H.sub[s_];
            // p = dq x + eq u + fqprev z  (src/ACME.jl:678-686)
            for (int r = 0; r < s.np; ++r) {
                double acc = 0.0;
                for (int j = 0; j < H.nx; ++j) acc = fma(c.M[s.o_dq + j * s.np + r], c.W(H.w_x + j), acc);
                for (int k = 0; k < H.nu; ++k) acc = fma(c.M[s.o_eq + k * s.np + r], un[k], acc);
                if (s_ > 0)
                    for (int k = 0; k < s.zoff; ++k) acc = fma(c.M[s.o_fqprev + k * s.np + r], c.W(H.w_z + k), acc);
                c.W(H.w_p + r) = acc;
            }
            int its;
            const bool conv = gen_homotopy_solve(c, s, its);
            its_sample += its;
            if (!conv) {                 // the policy of step! (src/ACME.jl:688-694)
                bool zfinite = true;
                for (int r = 0; r < s.nn; ++r) zfinite = zfinite && c.W(H.w_zz + r) * 0.0 == 0.0;
                if (zfinite) {
                    rep[RW_NWARN] += 1;
                    if (rep[RW_FIRST_NONCONV] < 0) rep[RW_FIRST_NONCONV] = A.sample_base + n;
                } else {
                    if (rep[RW_FIRST_NONFINITE] < 0) rep[RW_FIRST_NONFINITE] = A.sample_base + n;
                    dead = true;
                }
            }
            for (int r = 0; r < s.nn; ++r) c.W(H.w_z + s.zoff + r) = c.W(H.w_zz + r);
        }
        it_total += its_sample;
        if (its_sample > it_max) it_max = its_sample;
        if (dead) {
            for (int k = 0; k < H.ny; ++k) yn[k] = (double)NAN;
            continue;
        }
        // y = y0 + dy x + ey u + fy z (old x, :699-706);  x = x0 + a x + b u + c z (:708-714)
        for (int r = 0; r < H.ny; ++r) {
            double acc = c.M[H.o_y0 + r];
            for (int j = 0; j < H.nx; ++j) acc = fma(c.M[H.o_dy + j * H.ny + r], c.W(H.w_x + j), acc);
            for (int k = 0; k < H.nu; ++k) acc = fma(c.M[H.o_ey + k * H.ny + r], un[k], acc);
            for (int k = 0; k < H.nnt; ++k) acc = fma(c.M[H.o_fy + k * H.ny + r], c.W(H.w_z + k), acc);
            yn[r] = acc;
        }
        for (int r = 0; r < H.nx; ++r) {
            double acc = c.M[H.o_x0 + r];
            for (int j = 0; j < H.nx; ++j) acc = fma(c.M[H.o_a + j * H.nx + r], c.W(H.w_x + j), acc);
            for (int k = 0; k < H.nu; ++k) acc = fma(c.M[H.o_b + k * H.nx + r], un[k], acc);
            for (int k = 0; k < H.nnt; ++k) acc = fma(c.M[H.o_c + k * H.nx + r], c.W(H.w_z + k), acc);
            c.W(H.w_xn + r) = acc;
        }
        gen_copy(c, H.w_x, H.w_xn, H.nx);
    }
    for (int k = 0; k < H.nx; ++k) st[k] = c.W(H.w_x + k);
    for (int s_ = 0; s_ < H.nsub; ++s_) {
        const GenSub &s = H.sub[s_];
        for (int j = 0; j < s.np; ++j) st[H.nx + s.poff + j] = c.W(s.w_lp + j);
        for (int r = 0; r < s.nn; ++r) st[H.nx + H.npt + s.zoff + r] = c.W(s.w_lz + r);
    }
    rep[RW_ITERS_TOTAL] += it_total;
    if (it_max > rep[RW_ITERS_MAX]) rep[RW_ITERS_MAX] = it_max;
}
#endif  // ACME_DEV

}  // namespace acme
