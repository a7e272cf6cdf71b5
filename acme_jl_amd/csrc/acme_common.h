// acme_common.h -- POD types shared by the host packing code and the device kernels.
//
// Data layout of one circuit model in HBM/LDS ("model image"): every matrix of the
// reference's DiscreteModel (src/ACME.jl:118-148) stored column-major and back to back,
// exactly as Julia holds them, so that lane r of a 16-lane group reads row r of a matrix
// at consecutive addresses (bank-conflict free) for a fixed column.
#pragma once

#ifndef ACME_HD
#if defined(__HIPCC__)
#define ACME_HD __host__ __device__
#else
#define ACME_HD
#endif
#endif

namespace acme {

constexpr int GROUP = 16;         // lanes per circuit instance = one DPP row
constexpr int GROUPS_PER_WAVE = 4;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int INST_PER_BLOCK = GROUPS_PER_WAVE * WAVES_PER_BLOCK;
constexpr int CHUNK = 16;         // samples staged per coalesced u/y transfer
constexpr int ROWC = 38;          // precomputed constants per residual row
// rows of the common kinds (diode, Ebers-Moll BJT, potentiometer, padding) in the branch-free
// "unified row" form the non-RARE kernels evaluate (acme_kernel.h eval_row_unified):
//   xA = sA*e0, xB = sB*e1, w = w0 + w1*e2
//   res = cA*(exp(xA)-1) + cB*(exp(xB)-1) + g0*e0 + g1*e1 + g2*e2 + h*w*e1
//   dres/de0 = dA*exp(xA) + g0,  dres/de1 = dB*exp(xB) + g1 + h*w,  dres/de2 = g2 + h*e1
enum UnifiedRowConst { UR_SA = 24, UR_SB, UR_CA, UR_CB, UR_DA, UR_DB, UR_H, UR_SPARE,
                       UR_G0 = 32, UR_G1, UR_G2, UR_W0, UR_W1 };
constexpr int ROWI = 8;           // ints per residual row: kind, erow, flags, tc[0..3], (spare)
constexpr int MAX_NN = 16, MAX_NP = 16, MAX_NY = 16, MAX_NX = 32, MAX_NQ = 32, MAX_NU = 8, MAX_NSUB = 8;

// residual-row kinds (element kind of the row's element; same numbering as the element
// kinds of include/acme_hip.h, plus PAD for rows added by host-side shape padding)
enum RowKind { RK_NONE = 0, RK_DIODE = 1, RK_BJT = 2, RK_POT = 3, RK_MOSFET = 4, RK_MACAK = 5,
               RK_JA = 6, RK_PAD = 7 };
// row flags
enum RowFlags { RF_EARLY = 1, RF_KNEE = 2, RF_ILE = 4, RF_ILC = 8, RF_ETAEL = 16, RF_ETACL = 32 };

enum SolverKind { SOLVER_SIMPLE = 0, SOLVER_HOMOTOPY = 1, SOLVER_CACHING_HOMOTOPY = 2 };
// CachingSolver on the GPU: per instance and sub-problem the last CACHE stored solutions (p, z),
// first in first out (the reference keeps every stored solution in a k-d tree).  The p's -- what
// every lookup scans, one entry per lane -- live in LDS during a launch; the z's, read only on a
// hit, stay in HBM.  HBM layout per sub-problem: cp[np][CACHE] | count, head | cz[CACHE][nn] (entry-major: the nn lanes of
// an instance read / write one contiguous line).
constexpr int CACHE = 16;

struct Dims {
    int nn, nq, np, nx, nu, ny;
    int rare;  // shape compiled with / model needs the MOSFET, tanh op-amp, JA kinds
    int nsub;  // nonlinear sub-problems (shape: capacity; model: actual count)
    // residual rows that are LINEAR in z for a given p -- potentiometer halves whose position comes from an input
    // only (src/elements.jl:25-30) -- and are condensed out of the Newton system (acme_kernel.h "condensed solve";
    // shape: the kernel is built for exactly this many; 0: none)
    int nl;
};

// offsets (in doubles) of each matrix inside a model image
// Model image = [shared part][sub-problem 0][sub-problem 1]...  Every sub-problem is padded to
// the shape's (nn, nq, np); the unknowns of sub-problem s occupy z columns s*nn .. s*nn+nn-1.
// pexps/fqs/q0s are stored ROW-GATHERED: for residual row r and its t-th Jq non-zero (q row
// tc[r][t]) the image holds  fqr[(t*nn + j)*16 + r] = fq[tc[r][t], j]  (likewise pexpr, q0r).
// The lane that evaluates row r then reads consecutive addresses for fixed (t, j): no
// indirection and no LDS bank conflicts (the plain fq[tc + j*nq] form was ~39 % conflicts).
// On the big shapes (nn >= 8) columns are stored in PAIRS (Layout::gat): entries (t, j) and (t, j + 1),
// j even, of one row sit next to each other, so that two of them come with ONE 16-byte LDS read per
// lane (+5.4 % on the headline; the small shapes, with odd column counts to pad and few reads to
// save, lose 0.2 ... 2.5 % and keep single columns) (ds_read_b128:
// 3.5 ns per element with every SIMD busy against 6.7 ns for the ds_read2_b64 the compiler otherwise
// merges neighbouring columns into; tools/ubench/ldsread.hip).
struct Layout {
    int a, b, c, x0, dy, ey, fy, y0;            // shared, absolute offsets (dy.. = a.. + nx: extra rows)
    int ld;                                     // leading dimension of those matrices: nx + ny
    int sub0, sub_stride;                       // first sub-problem block, distance to the next
    int dq, eq, fqprev, pexpr, fqr, q0r;        // offsets relative to a sub-problem block
    int total;
    int pairs;                                  // row-gathered copies stored in column pairs (single-sub-problem shapes with nn >= 8)
    // rows per column of the row-gathered copies: nn + 1 where that is less than the 16 lanes -- the
    // residual rows and one all-zero row, which the lanes beyond nn read (Monte-Carlo batches keep 16
    // private images per block in LDS: the copies are two thirds of an image)
    int gs;
    // index (relative to pexpr / fqr) of the row-gathered entry (t, j) of row r, n columns per term
    ACME_HD constexpr int gat(int t, int j, int r, int n) const {
        return pairs ? ((t * ((n + 1) / 2) + j / 2) * GROUP + r) * 2 + (j & 1) : (t * n + j) * gs + r;
    }
    ACME_HD constexpr int gat_size(int nt, int n) const { return pairs ? nt * ((n + 1) / 2) * 2 * GROUP : nt * n * gs; }
    ACME_HD constexpr int q0i(int t, int r) const { return q0r + t * gs + r; }
    // The matrices of the linear update, [x0 | a | b | c] over [y0 | dy | ey | fy] (ld = nx + ny rows):
    // image index of row `row` of combined column `col` (0: x0/y0, then the nx columns of a/dy, the nu
    // of b/ey, the nz of c/fy).  linp: stored in column PAIRS at lin0 like the row-gathered copies
    // (the one-pass update of the big shapes reads two columns per ds_read_b128); otherwise the
    // separate column-major blocks a, b, c, x0.
    int linp, lin0;
    // [dq | eq] of a sub-problem block (np rows, nx + nu columns): column pairs at dq when `pairs`
    ACME_HD constexpr int pq(int col, int row, int np, int nx) const {
        if (pairs) return dq + ((col / 2) * GROUP + row) * 2 + (col & 1);
        return col < nx ? dq + col * np + row : eq + (col - nx) * np + row;
    }
    ACME_HD constexpr int lin(int col, int row, int nx, int nu) const {
        if (linp) return lin0 + ((col / 2) * GROUP + row) * 2 + (col & 1);
        if (col == 0) return x0 + row;
        col -= 1;
        if (col < nx) return a + col * ld + row;
        col -= nx;
        if (col < nu) return b + col * ld + row;
        return c + (col - nu) * ld + row;
    }
};

ACME_HD constexpr Layout make_layout(int nn, int nq, int np, int nx, int nu, int ny, int nt, int nsub) {
    Layout L{};
    (void)nq;
    const int nz = nsub * nn;
    // [a; dy], [b; ey], [c; fy], [x0; y0] are stored as single matrices with nx + ny rows (leading
    // dimension ld): when nx + ny <= 16 one pass over 16 lanes yields the new state AND the output
    const int ld = nx + ny;
    int o = 0;
    L.ld = ld;
    L.pairs = (nn >= 8 && nsub == 1) ? 1 : 0;     // (the 4-sub-problem shapes have no LDS to spare for the padding)
    L.gs = (L.pairs || nn + 1 >= GROUP) ? GROUP : nn + 1;
    L.linp = (L.pairs && ld <= GROUP && nx > 0) ? 1 : 0;     // (on the small shapes: +-0, measured)
    if (L.linp) {
        L.lin0 = o; o += ((1 + nx + nu + nz + 1) / 2) * 2 * GROUP;
    }
    L.a = o;    o += L.linp ? 0 : ld * nx;
    L.b = o;    o += L.linp ? 0 : ld * nu;
    L.c = o;    o += L.linp ? 0 : ld * nz;
    L.x0 = o;   o += L.linp ? 0 : ld;
    L.dy = L.a + nx;
    L.ey = L.b + nx;
    L.fy = L.c + nx;
    L.y0 = L.x0 + nx;
    L.sub0 = (o + 1) & ~1;
    int r = 0;
    L.dq = r;     r += L.pairs ? ((nx + nu + 1) / 2) * 2 * GROUP : np * nx;   // see Layout::pq
    L.eq = r;     r += L.pairs ? 0 : np * nu;
    L.fqprev = r; r += nsub > 1 ? np * nz : 0;   // only read by sub-problems after the first
    r = (r + 1) & ~1;                                        // pairs start 16-byte aligned
    L.pexpr = r;  r += L.gat_size(nt, np);                   // see Layout::gat
    L.fqr = r;    r += L.gat_size(nt, nn);
    L.q0r = r;    r += nt * L.gs;
    L.sub_stride = (r + 1) & ~1;
    // tail padding: lanes beyond a matrix's row count read (finite) neighbours, never past
    // the end of the image
    o = L.sub0 + nsub * L.sub_stride + 2 * GROUP;
    L.total = (o + 1) & ~1;
    return L;
}

// Constant block of the lane-per-instance kernel (acme_lane_kernel.h): the same numbers as the model
// image, regrouped so that everything one residual row (or one row of the linear update) needs is
// CONTIGUOUS -- the kernel reads it with wave-uniform addresses, i.e. wide scalar loads.
//   residual row r at r * row:  q0r[3] | pexpr[3][np] | fqr[3][nn] | unified row constants UR_SA..UR_W1 | kind
//   p row i at p0 + i * pstr:   dq[i][0..nx) | eq[i][0..nu)
//   y row i at y0 + i * xstr:   y0[i] | dy[i][0..nx) | ey[i][0..nu) | fy[i][0..nn)
//   x row i at x0 + i * xstr:   x0[i] | a[i][0..nx)  | b[i][0..nu)  | c[i][0..nn)
struct LaneLayout {
    int row, q0, pexp, fq, ur, kind;    // row stride and offsets inside a residual row
    int p0, pstr, y0, x0, xstr, total;
};
ACME_HD constexpr LaneLayout make_lane_layout(int nn, int np, int nx, int nu, int ny) {
    LaneLayout l{};
    l.q0 = 0;
    l.pexp = 3;
    l.fq = 3 + 3 * np;
    l.ur = l.fq + 3 * nn;
    l.kind = l.ur + (UR_W1 - UR_SA + 1);
    l.row = (l.kind + 1 + 7) & ~7;
    l.pstr = (nx + nu + 1) & ~1;
    l.xstr = (1 + nx + nu + nn + 1) & ~1;
    l.p0 = nn * l.row;
    l.y0 = l.p0 + np * l.pstr;
    l.x0 = l.y0 + ny * l.xstr;
    l.total = ((l.x0 + nx * l.xstr) + 7) & ~7;
    return l;
}

// per-instance report (int64 words), mirrors the reference's failure semantics
// (src/ACME.jl:688-694)
enum ReportWord { RW_NWARN = 0, RW_FIRST_NONCONV = 1, RW_FIRST_NONFINITE = 2, RW_ITERS_TOTAL = 3,
                  RW_ITERS_MAX = 4, RW_WORDS = 5 };

struct KArgs {
    const double *image;     // model image(s)
    long long image_stride;  // 0: one shared image; else doubles between per-instance images
    const double *rowc;      // ROWC x 16 row constants (const index major)
    const int *rowi;         // ROWI x 16 row ints
    // Per-instance ELEMENT TABLES (every model of the reference carries its own element closures, src/elements.jl:236-245,
    // 309-406: a sweep over a diode's is or a transistor's beta): doubles / ints between the tables of consecutive
    // instances in rowc / rowi; 0: one shared table.  A block then stages its 16 instances' tables in LDS, as it does
    // their images.
    long long table_stride, tablei_stride;
    const double *lanec;     // constant block of the lane-per-instance kernel (LaneLayout), or nullptr
    const double *u;         // [n_inst][T][nu_io]
    double *y;               // [n_inst][T][ny_io]
    double *state;           // [n_inst][nx + np + nn] : x | last_p | last_z
    long long *report;       // [n_inst][RW_WORDS]
    double *cache;           // [n_inst][nsub_shape][cache_doubles]: the solution caches between launches
    int *roworder;           // [n_inst][nsub_shape][16]: lane -> residual row assignment the lanes had
                             // adopted when the previous launch ended (identity = the host hint); kept
                             // so that a run split over several launches repeats the one-launch arithmetic
    long long n_inst;
    long long T;
    long long sample_base;   // global index of the first sample of this launch
    double tol;
    int maxiter;
    int solver;
    int nu_io, ny_io;        // strides of u / y in HBM (<= template NU / NY)
    int nterms;              // max non-zeros per Jq row (2..4)
    int has_bjt;             // any BJT row -> second exp needed
    int rare_kinds;          // any MOSFET/MACAK/JA row
    // solver-plugin mode (acme_batch_solve): if p_in != nullptr the launch performs ONE
    // solve(solver, p) per instance instead of running samples: p comes from p_in, the
    // solution / hasconverged / needediterations go to z_out / conv_out / iters_out, x is
    // not touched, the extrapolation origin is updated as by any solve
    const double *p_in;      // [n_inst][np_io]
    double *z_out;           // [n_inst][nn_io]
    int *conv_out, *iters_out;
    int np_io, nn_io;
    double *jac_out;         // MODE_JAC kernels: [n_inst][np_io][nn_io] = -(J \ Jp) at the origin of solve_sub
    int solve_sub;           // sub-problem acme_batch_solve / the Jacobian export addresses
    int nsub;                // actual number of sub-problems (<= the shape's NSUB)
    // condensed shapes (Dims::nl > 0) keep z in a PERMUTED basis (the linear rows' pivot columns first, acme_pack.h):
    // zperm[i] = the caller's index of the kernel's z_i (identity otherwise); z_out / jac_out are written through it
    int zperm[GROUP];
    // A launch over a SUBSET of the batch's instances (isolation of slow instances, acme_batch_set_isolation): n_inst
    // is then the number of instances of the launch and inst_map[slot] the batch instance slot `slot` works on
    // (nullptr: slot i is instance i)
    const int *inst_map;
    // lane-per-instance kernel: instances per wave (1 .. 64; the lanes beyond idle).  A batch that leaves SIMDs empty
    // is spread thinner: a wave's Newton loop runs as long as its slowest lane needs, and fewer lanes have a smaller maximum
    int lane_density;
    int *cflags;             // [n_inst]: bit 0 = the extrapolation origin may lie OFF the linear rows' subspace
                             // (initial solution, acme_batch_set_state, an iterate accepted without a Newton step)
    // samples between the rows of consecutive instances in u / y (0: T -- the launch covers whole rows).  A host-buffer
    // run reads or writes a time slice of the caller's arrays in place: rows T_total apart, T of them per launch
    long long u_stride, y_stride;
    // streamed host run: u is being copied into HBM while the kernel runs; *u_ready (host memory, written by the host
    // after each chunk of the copy has landed) = number of samples of every row that are there.  A wave that gets ahead
    // of the copy waits.  nullptr: all of u is there.
    const long long *u_ready;
};

}  // namespace acme
