// acme_pack.h -- host-side packing of a DiscreteModel into the device "model image"
// plus the per-row element tables (pure C++, no HIP; shared by the HIP library and the
// wave emulator used in CPU tests).
//
// The reference passes the nonlinearity as Julia closures (ParametricNonLinEq,
// src/solvers.jl:6-36; CircuitNLFunc, src/circuit.jl:6-20,68-86).  Closures cannot cross
// a C ABI, so the model carries an element table instead; here it is flattened to one
// descriptor per residual row (= per lane) with every loop-invariant sub-expression of
// the element functions (src/elements.jl) pre-evaluated in the same operation order.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "acme_common.h"
#include "acme_generic.h"
#include "acme_balance.h"
#include "acme_shapes.h"

namespace acme {

constexpr int MAX_ELEM_PAR = 16;
enum ElemKind { EK_DIODE = 1, EK_BJT = 2, EK_POT = 3, EK_MOSFET = 4, EK_MACAK = 5, EK_JA = 6 };

struct HostSub {
    int nn = 0, nq = 0, np = 0;
    std::vector<double> pexp, dq, eq, fqprev, fq, q0, init_z;
    std::vector<int> kind, qoff, roff;
    std::vector<double> par;  // n_elems x MAX_ELEM_PAR
    // performance hint: row_order[pos] = residual row evaluated by lane `pos` (empty = natural
    // order).  Pre-ordering the equations in the usual pivot order makes the partially
    // pivoted LU (src/solvers.jl:58-78) find its pivot in place almost always, i.e. without
    // the cross-lane row interchange; pivots and arithmetic are unchanged.
    std::vector<int> row_order;
};

struct HostModel {
    int nx = 0, nu = 0, ny = 0, nn_total = 0;
    std::vector<double> a, b, c, x0, dy, ey, fy, y0;
    std::vector<HostSub> subs;
};

struct Packed {
    Dims shape{};   // instantiated kernel shape
    Dims actual{};  // the model's own dimensions
    std::vector<double> image, rowc, init_state;
    std::vector<double> lanec;   // LaneLayout block (models with one nonlinear sub-problem)
    std::vector<int> rowi;
    int nterms = 2, has_bjt = 0, rare_kinds = 0;
    // condensed shapes (Dims::nl > 0): the kernel's z basis is a permutation of the caller's --
    // zperm[i] = caller's index of the kernel's z_i, zinv its inverse (identity otherwise) -- and the
    // lanes' residual rows start with the linear ones (lrows, in pivot order)
    int zperm[GROUP], zinv[GROUP];
    std::vector<int> lrows, lcols;
};

// The condensation plan of a sub-problem: its residual rows that are linear in z for a given p (potentiometer
// halves, res = v - r w i with w = pos or 1 - pos, src/elements.jl:25-30, whose `pos` entry of q has an all-zero
// fq row: pos then comes from the inputs / states alone, src/ACME.jl:176-189) and ONE static pivot column of
// z for each -- chosen by complete pivoting on the row-equilibrated A_L = fq[v] - r w fq[i] at mid travel and
// accepted only if A_LL = A_L[:, lcols] stays well conditioned over the whole travel of every pot
// (equilibrated condition number <= 1e8 on a 3-point grid per pot, both end stops pulled in by 2 %: a pot AT an
// end stop can short a branch and make the model itself singular, e.g. the drive pot of examples/superover.jl
// at 1.0).  With the rows in lanes 0 .. nl-1 and the columns first, the kernel eliminates them once per change
// of the pot positions instead of once per Newton iteration (acme_kernel.h: "condensed solve").
struct CondPlan {
    std::vector<int> lrows, lcols;      // residual rows / z columns, in pivot order
};
// forced: vet THIS model against a plan made for another (a batch's per-instance models take the batch model's plan, so
// that they share its lane assignment): the same potentiometer rows must be linear here too (their `pos` driven by the
// inputs alone) and A_LL with the plan's pivot columns well conditioned over the pots' travel for THIS model's
// component values -- an instance with other resistor values is not vouched for by the batch model's check.
inline bool plan_condense(const HostSub &s, CondPlan &plan, const CondPlan *forced = nullptr) {
    struct LinRow { int row, v, i; double r, w0, w1; int pot; };
    std::vector<LinRow> rows;
    int npots = 0;
    for (size_t e = 0; e < s.kind.size(); ++e) {
        if (s.kind[e] != EK_POT) continue;
        const int q0 = s.qoff[e];
        bool pos_free = true;
        for (int j = 0; j < s.nn; ++j)
            if (s.fq[(size_t)j * s.nq + q0 + 4] != 0.0) pos_free = false;
        if (!pos_free) continue;
        const double r = s.par[e * MAX_ELEM_PAR];
        rows.push_back({s.roff[e] + 0, q0 + 0, q0 + 2, r, 0.0, 1.0, npots});
        rows.push_back({s.roff[e] + 1, q0 + 1, q0 + 3, r, 1.0, -1.0, npots});
        ++npots;
    }
    const int nl = (int)rows.size(), nn = s.nn;
    plan.lrows.clear();
    plan.lcols.clear();
    if (nl == 0 || nl >= nn || npots > 8) return false;
    auto build = [&](const std::vector<double> &pos, std::vector<double> &A) {     // nl x nn, row-major, equilibrated rows
        A.assign((size_t)nl * nn, 0.0);
        for (int k = 0; k < nl; ++k) {
            const LinRow &R = rows[k];
            const double rw = R.r * (R.w0 + R.w1 * pos[R.pot]);
            double mx = 0.0;
            for (int j = 0; j < nn; ++j) {
                A[(size_t)k * nn + j] = s.fq[(size_t)j * s.nq + R.v] - rw * s.fq[(size_t)j * s.nq + R.i];
                mx = std::fmax(mx, std::fabs(A[(size_t)k * nn + j]));
            }
            if (mx > 0.0) for (int j = 0; j < nn; ++j) A[(size_t)k * nn + j] /= mx;
        }
    };
    std::vector<double> A, pos(npots, 0.5);
    build(pos, A);
    std::vector<int> rleft(nl), cleft(nn);
    for (int k = 0; k < nl; ++k) rleft[k] = k;
    for (int j = 0; j < nn; ++j) cleft[j] = j;
    std::vector<int> prow, pcol;
    if (forced) {
        if ((int)forced->lrows.size() != nl || (int)forced->lcols.size() != nl) return false;
        for (int k = 0; k < nl; ++k) {
            int idx = -1;
            for (int r = 0; r < nl; ++r)
                if (rows[r].row == forced->lrows[k]) idx = r;
            if (idx < 0 || forced->lcols[k] < 0 || forced->lcols[k] >= nn) return false;
            prow.push_back(idx);
            pcol.push_back(forced->lcols[k]);
        }
    }
    for (int step = 0; step < (forced ? 0 : nl); ++step) {
        double best = 0.0;
        int br = -1, bc = -1;
        for (int r : rleft)
            for (int c : cleft)
                if (std::fabs(A[(size_t)r * nn + c]) > best) { best = std::fabs(A[(size_t)r * nn + c]); br = r; bc = c; }
        if (br < 0 || best < 1e-12) return false;       // the linear rows are dependent at mid travel
        prow.push_back(br);
        pcol.push_back(bc);
        rleft.erase(std::find(rleft.begin(), rleft.end(), br));
        cleft.erase(std::find(cleft.begin(), cleft.end(), bc));
        for (int r : rleft) {
            const double l = A[(size_t)r * nn + bc] / A[(size_t)br * nn + bc];
            for (int c = 0; c < nn; ++c) A[(size_t)r * nn + c] -= l * A[(size_t)br * nn + c];
        }
    }
    // conditioning of A_LL over the pots' travel
    const double grid[3] = {0.02, 0.5, 0.98};
    long combos = 1;
    for (int k = 0; k < npots; ++k) combos *= 3;
    for (long c = 0; c < combos; ++c) {
        long cc = c;
        for (int k = 0; k < npots; ++k) { pos[k] = grid[cc % 3]; cc /= 3; }
        build(pos, A);
        std::vector<double> B((size_t)nl * nl), Binv((size_t)nl * nl, 0.0);
        for (int i = 0; i < nl; ++i) {
            double mx = 0.0;
            for (int j = 0; j < nl; ++j) mx = std::fmax(mx, std::fabs(A[(size_t)prow[i] * nn + pcol[j]]));
            if (mx == 0.0) return false;
            for (int j = 0; j < nl; ++j) B[(size_t)i * nl + j] = A[(size_t)prow[i] * nn + pcol[j]] / mx;
            Binv[(size_t)i * nl + i] = 1.0;
        }
        double nB = 0.0;
        for (int i = 0; i < nl; ++i) {
            double rs = 0.0;
            for (int j = 0; j < nl; ++j) rs += std::fabs(B[(size_t)i * nl + j]);
            nB = std::fmax(nB, rs);
        }
        for (int k = 0; k < nl; ++k) {       // Gauss-Jordan with partial pivoting on [B | I]
            int pr = k;
            for (int i = k + 1; i < nl; ++i)
                if (std::fabs(B[(size_t)i * nl + k]) > std::fabs(B[(size_t)pr * nl + k])) pr = i;
            if (B[(size_t)pr * nl + k] == 0.0) return false;
            if (pr != k)
                for (int j = 0; j < nl; ++j) {
                    std::swap(B[(size_t)k * nl + j], B[(size_t)pr * nl + j]);
                    std::swap(Binv[(size_t)k * nl + j], Binv[(size_t)pr * nl + j]);
                }
            const double inv = 1.0 / B[(size_t)k * nl + k];
            for (int j = 0; j < nl; ++j) { B[(size_t)k * nl + j] *= inv; Binv[(size_t)k * nl + j] *= inv; }
            for (int i = 0; i < nl; ++i) {
                if (i == k) continue;
                const double l = B[(size_t)i * nl + k];
                for (int j = 0; j < nl; ++j) {
                    B[(size_t)i * nl + j] -= l * B[(size_t)k * nl + j];
                    Binv[(size_t)i * nl + j] -= l * Binv[(size_t)k * nl + j];
                }
            }
        }
        double nI = 0.0;
        for (int i = 0; i < nl; ++i) {
            double rs = 0.0;
            for (int j = 0; j < nl; ++j) rs += std::fabs(Binv[(size_t)i * nl + j]);
            nI = std::fmax(nI, rs);
        }
        if (!(nB * nI <= 1e8)) return false;
    }
    for (int k = 0; k < nl; ++k) {
        plan.lrows.push_back(rows[prow[k]].row);
        plan.lcols.push_back(pcol[k]);
    }
    return true;
}
// the model with the z basis of its (single) sub-problem permuted: kernel column i = caller's column zperm[i]
inline void permute_z(HostModel &m, const int *zperm) {
    HostSub &s = m.subs[0];
    const int nn = s.nn;
    std::vector<double> fq(s.fq.size()), iz(nn), c(m.c.size()), fy(m.fy.size());
    for (int j = 0; j < nn; ++j) {
        for (int i = 0; i < s.nq; ++i) fq[(size_t)j * s.nq + i] = s.fq[(size_t)zperm[j] * s.nq + i];
        iz[j] = s.init_z[zperm[j]];
        for (int i = 0; i < m.nx; ++i) c[(size_t)j * m.nx + i] = m.c[(size_t)zperm[j] * m.nx + i];
        for (int i = 0; i < m.ny; ++i) fy[(size_t)j * m.ny + i] = m.fy[(size_t)zperm[j] * m.ny + i];
    }
    s.fq = fq; s.init_z = iz; m.c = c; m.fy = fy;
}

inline const std::vector<Dims> &shape_list() {
    static const std::vector<Dims> v = {
#define ACME_X(nn, nq, np, nx, nu, ny, rare, nsub, nl) Dims{nn, nq, np, nx, nu, ny, rare, (nn) > 0 ? (nsub) : 0, nl},
        ACME_SHAPES(ACME_X)
#undef ACME_X
    };
    return v;
}

inline void kind_shape(int kind, int &nq, int &nn) {
    switch (kind) {
    case EK_DIODE: nq = 2; nn = 1; break;
    case EK_BJT: nq = 4; nn = 2; break;
    case EK_POT: nq = 5; nn = 2; break;
    case EK_MOSFET: nq = 3; nn = 1; break;
    case EK_MACAK: nq = 2; nn = 1; break;
    case EK_JA: nq = 4; nn = 1; break;
    default: nq = 0; nn = 0; break;
    }
}

// pick the instantiated shape a model runs in: exact match first, else the cheapest
// (least padded work) shape that contains it
inline bool choose_shape(const Dims &d, Dims &out) {
    const auto &list = shape_list();
    // linear models prefer the solver-less shapes (pass 0); anything else, or a linear model
    // too big for them, takes the cheapest shape that contains it (pass 1)
    for (int pass = (d.nn == 0 ? 0 : 1); pass < 2; ++pass) {
        long best = -1;
        for (const Dims &s : list) {
            if (d.rare && !s.rare) continue;
            if (d.nsub > s.nsub) continue;
            if (pass == 0 && s.nn != 0) continue;
            if (s.nn == d.nn && s.nq == d.nq && s.np == d.np && s.nx == d.nx && s.nu == d.nu && s.ny == d.ny &&
                s.nsub == d.nsub && s.nl == d.nl) {
                out = s;
                return true;
            }
            if (s.nl != 0) continue;     // a condensed kernel is built for exactly its model
            bool fits = d.nn <= s.nn && d.np <= s.np && d.nx <= s.nx && d.nu <= s.nu && d.ny <= s.ny &&
                        d.nq + (s.nn - d.nn) <= s.nq;
            if (!fits) continue;
            long cost = (long)(s.nsub ? s.nsub : 1) * ((long)s.nn * s.nn * s.nn + (long)s.nq * (s.nn + s.np)) +
                        (long)(s.nx + s.np + s.ny) * (s.nx + s.nu + s.nn);
            if (best < 0 || cost < best) {
                best = cost;
                out = s;
            }
        }
        if (best >= 0) return true;
    }
    return false;
}

inline void put(std::vector<double> &img, int off, int ld, const std::vector<double> &m, int r, int c) {
    // copy an r x c column-major matrix into a column-major block with leading dim ld
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < r; ++i) img[off + (size_t)j * ld + i] = m[(size_t)j * r + i];
}

// constants, flags and Jq-column table of the residual rows of one element
inline bool describe_element(int kind, const double *p, int er, int q0_, double *rc /*[ROWC]*/, int *ri /*[ROWI]*/,
                             Packed &P, std::string &err) {
    for (int c = 0; c < ROWC; ++c) rc[c] = 0.0;
    for (int w = 0; w < ROWI; ++w) ri[w] = 0;
    ri[0] = kind;  // RowKind numbering == element kind numbering
    ri[1] = er;
    // q rows the residual row depends on = columns of its Jq non-zeros
    int tcs[4] = {q0_, q0_, q0_, q0_};
    int flags = 0;
    switch (kind) {
    case EK_DIODE: {  // src/elements.jl:238-244 -- (v, i)
        double is = p[0], eta = p[1];
        tcs[1] = q0_ + 1;
        rc[0] = 1 / (25e-3 * eta);
        rc[1] = is;
        rc[2] = is / (25e-3 * eta);
        // res = is*(exp(v/(eta*vT)) - 1) - i
        rc[UR_SA] = rc[0]; rc[UR_CA] = rc[1]; rc[UR_DA] = rc[2]; rc[UR_G1] = -1.0;
        break;
    }
    case EK_BJT: {  // src/elements.jl:323-401 -- (vE, vC, iE|iC)
        double ise = p[0], isc = p[1], etae = p[2], etac = p[3], bf = p[4], br = p[5], ile = p[6], ilc = p[7],
               etael = p[8], etacl = p[9], vaf = p[10], var = p[11], ikf = p[12], ikr = p[13];
        tcs[1] = q0_ + 1;
        tcs[2] = q0_ + 2 + er;
        rc[0] = 1 / (25e-3 * etae);
        rc[1] = 1 / (25e-3 * etac);
        rc[2] = bf / (1 + bf) * ise;
        rc[3] = br / (1 + br) * isc;
        rc[4] = bf / (1 + bf) * ise / (25e-3 * etae);
        rc[5] = br / (1 + br) * isc / (25e-3 * etac);
        rc[6] = 1 / bf;
        rc[7] = 1 / br;
        rc[8] = 1 / var;
        rc[9] = 1 / vaf;
        rc[10] = 1 / ikf;
        rc[11] = 1 / ikr;
        rc[12] = ile;
        rc[13] = ilc;
        rc[14] = 1 / (25e-3 * etael);
        rc[15] = 1 / (25e-3 * etacl);
        rc[16] = ile / (25e-3 * etae);  // sic: the reference uses eta_e here (:384)
        rc[17] = ilc / (25e-3 * etac);  // and eta_c here (:395)
        rc[18] = -1 / var;
        rc[19] = -1 / vaf;
        if (!(std::isinf(var) && std::isinf(vaf))) flags |= RF_EARLY;
        if (!(std::isinf(ikf) && std::isinf(ikr))) flags |= RF_KNEE;
        if (ile != 0) flags |= RF_ILE;
        if (ilc != 0) flags |= RF_ILC;
        if (etael != etae) flags |= RF_ETAEL;
        if (etacl != etac) flags |= RF_ETACL;
        P.has_bjt = 1;
        if (P.nterms < 3) P.nterms = 3;
        if (flags) P.rare_kinds = 1;  // Gummel-Poon terms live in the RARE build
        // Ebers-Moll: row 0  (i_f - i_r) + i_f/bf - iE,   row 1  -(i_f - i_r) + i_r/br - iC
        rc[UR_SA] = rc[0]; rc[UR_SB] = rc[1]; rc[UR_G2] = -1.0;
        if (er == 0) {
            rc[UR_CA] = rc[2] * (1 + rc[6]); rc[UR_CB] = -rc[3];
            rc[UR_DA] = rc[4] * (1 + rc[6]); rc[UR_DB] = -rc[5];
        } else {
            rc[UR_CA] = -rc[2]; rc[UR_CB] = rc[3] * (1 + rc[7]);
            rc[UR_DA] = -rc[4]; rc[UR_DB] = rc[5] * (1 + rc[7]);
        }
        break;
    }
    case EK_POT:  // src/elements.jl:25-30 -- (v, i, pos) of this half
        tcs[0] = q0_ + er;
        tcs[1] = q0_ + 2 + er;
        tcs[2] = q0_ + 4;
        rc[0] = p[0];
        if (P.nterms < 3) P.nterms = 3;
        // res = v - r*w*i with w = pos (first half) or 1 - pos (second half); the reference's
        // Jacobian has -r*i in the pos column of BOTH halves (src/elements.jl:28) -- kept
        rc[UR_G0] = 1.0; rc[UR_H] = -p[0];
        rc[UR_W0] = er == 0 ? 0.0 : 1.0; rc[UR_W1] = er == 0 ? 1.0 : -1.0;
        break;
    case EK_MOSFET: {  // src/elements.jl:444-479
        tcs[1] = q0_ + 1;
        tcs[2] = q0_ + 2;
        for (int c = 0; c < 12; ++c) rc[c] = p[c];
        int nvt = (int)p[2], na = (int)p[7];
        for (int k = 1; k < nvt; ++k) rc[12 + k - 1] = p[3 + k] * k;
        for (int k = 1; k < na; ++k) rc[15 + k - 1] = p[8 + k] * k;
        P.rare_kinds = 1;
        if (P.nterms < 3) P.nterms = 3;
        break;
    }
    case EK_MACAK:  // src/elements.jl:540-546
        tcs[1] = q0_ + 1;
        rc[0] = p[0];
        rc[1] = p[1];
        rc[2] = p[0] / p[1];
        P.rare_kinds = 1;
        break;
    case EK_JA: {  // src/elements.jl:107-129
        double Ms = p[0], a = p[1], alpha = p[2], c = p[3];
        tcs[1] = q0_ + 1;
        tcs[2] = q0_ + 2;
        tcs[3] = q0_ + 3;
        for (int k = 0; k < 5; ++k) rc[k] = p[k];
        rc[5] = 1e-4 / Ms;
        rc[6] = c * Ms / a;
        rc[7] = c * Ms / a * alpha;
        P.rare_kinds = 1;
        P.nterms = 4;
        break;
    }
    default:
        err = "unknown element kind";
        return false;
    }
    ri[2] = flags;
    for (int t = 0; t < 4; ++t) ri[3 + t] = tcs[t];
    return true;
}

inline bool pack_model(const HostModel &m_in, Packed &P, std::string &err, const Dims *force_shape = nullptr,
                       const Packed *force_plan = nullptr, bool allow_condense = true) {
    // Condensation (see CondPlan): decided here, once per model -- per-instance models of a batch take the batch
    // model's plan (force_plan) so that they share its element table and lane assignment.  ACME_CONDENSE=0 in the
    // environment keeps the plain kernel (A/B measurements, tests of the uncondensed path), and so does allow_condense =
    // false: a batch whose instances turn out to differ in their element parameters (acme_batch_set_matrices) is rebuilt
    // on the plain shape, the condensed kernels have no per-instance element tables.
    for (int i = 0; i < GROUP; ++i) P.zperm[i] = P.zinv[i] = i;
    P.lrows.clear();
    P.lcols.clear();
    HostModel m_perm;
    const HostModel *mp = &m_in;
    int nl_model = 0;
    {
        const char *e = getenv("ACME_CONDENSE");
        const bool allowed = allow_condense && !(e && e[0] == '0') && m_in.subs.size() == 1 && (!force_shape || force_shape->nl > 0);
        CondPlan plan;
        bool have = false;
        if (allowed && force_plan && !force_plan->lrows.empty()) {
            const CondPlan fp{force_plan->lrows, force_plan->lcols};
            if (!plan_condense(m_in.subs[0], plan, &fp)) {
                err = "the batch model's condensation of the potentiometer rows does not hold for this instance's component values "
                      "(a row is not linear here, or the linear block is ill conditioned): create the batch with ACME_CONDENSE=0";
                return false;
            }
            have = true;
        } else if (allowed && !force_plan) {
            have = plan_condense(m_in.subs[0], plan);
        }
        if (have) {     // is there a kernel built for it?
            const HostSub &s0 = m_in.subs[0];
            Dims probe{s0.nn, s0.nq, s0.np, m_in.nx, m_in.nu, m_in.ny, 0, 1, (int)plan.lrows.size()};
            bool found = false;
            for (const Dims &sh : shape_list())
                if (sh.nn == probe.nn && sh.nq == probe.nq && sh.np == probe.np && sh.nx == probe.nx && sh.nu == probe.nu &&
                    sh.ny == probe.ny && sh.nsub == 1 && sh.nl == probe.nl && !sh.rare)
                    found = true;
            if (force_shape && force_shape->nl != probe.nl) found = false;
            if (found) {
                nl_model = probe.nl;
                P.lrows = plan.lrows;
                P.lcols = plan.lcols;
                std::vector<char> used(s0.nn, 0);
                int k = 0;
                for (int c : plan.lcols) { P.zperm[k++] = c; used[c] = 1; }
                for (int c = 0; c < s0.nn; ++c) if (!used[c]) P.zperm[k++] = c;
                for (int i = 0; i < s0.nn; ++i) P.zinv[P.zperm[i]] = i;
                m_perm = m_in;
                permute_z(m_perm, P.zperm);
                // lanes: the linear rows first (pivot order), then the others in the order of the caller's hint
                std::vector<int> order = plan.lrows, rest = s0.row_order;
                if (rest.empty()) for (int r = 0; r < s0.nn; ++r) rest.push_back(r);
                for (int r : rest) if (std::find(plan.lrows.begin(), plan.lrows.end(), r) == plan.lrows.end()) order.push_back(r);
                m_perm.subs[0].row_order = order;
                mp = &m_perm;
            }
        }
    }
    if (force_shape && force_shape->nl != nl_model) {
        err = "model cannot be condensed like the batch's model (its potentiometer rows differ)";
        return false;
    }
    const HostModel &m = *mp;
    Dims d{};
    d.nl = nl_model;
    d.nx = m.nx; d.nu = m.nu; d.ny = m.ny;
    d.nsub = (int)m.subs.size();
    if (d.nsub > MAX_NSUB) {
        err = "more than 8 nonlinear sub-problems are not supported by the GPU path; derive the model with "
              "decompose_nonlinearity=false";
        return false;
    }
    // every sub-problem is padded to the largest one
    int pad_need = 0;
    for (const HostSub &s : m.subs) {
        if (s.nn > d.nn) d.nn = s.nn;
        if (s.np > d.np) d.np = s.np;
        for (size_t e = 0; e < s.kind.size(); ++e) {
            int kd = s.kind[e];
            if (kd == EK_MOSFET || kd == EK_MACAK || kd == EK_JA) d.rare = 1;
            if (kd == EK_BJT) {  // any Gummel-Poon refinement (src/elements.jl:331-396)
                const double *p = &s.par[e * MAX_ELEM_PAR];
                if (p[6] != 0 || p[7] != 0 || !std::isinf(p[10]) || !std::isinf(p[11]) || !std::isinf(p[12]) ||
                    !std::isinf(p[13]))
                    d.rare = 1;
            }
        }
    }
    for (const HostSub &s : m.subs) {  // q rows incl. the rows the nn-padding of that sub-problem adds
        int need = s.nq + (d.nn - s.nn);
        if (need > pad_need) pad_need = need;
    }
    d.nq = pad_need;
    if (d.nn > MAX_NN || d.nq > MAX_NQ || d.np > MAX_NP || d.nx > MAX_NX || d.nu > MAX_NU || d.ny > MAX_NY) {
        err = "model dimensions exceed the limits of the 16-lane kernel (nn<=16, nq<=32, np<=16, nx<=32, nu<=8, ny<=16)";
        return false;
    }
    Dims S{};
    auto fits = [&](const Dims &s_) {
        return d.nn <= s_.nn && d.np <= s_.np && d.nx <= s_.nx && d.nu <= s_.nu && d.ny <= s_.ny && d.nsub <= s_.nsub &&
               d.nq + (s_.nn - d.nn) <= s_.nq && (!d.rare || s_.rare) && s_.nl == d.nl;
    };
    if (force_shape) {
        S = *force_shape;
        if (!fits(S)) {
            err = "model does not fit the batch's kernel shape";
            return false;
        }
    } else if (!choose_shape(d, S)) {
        err = "no instantiated kernel shape contains this model";
        return false;
    }
    P.shape = S;
    P.actual = d;
    const int NT = S.rare ? 4 : 3;
    const int NSUBr = S.nsub > 0 ? S.nsub : 1;
    const Layout L = make_layout(S.nn, S.nq, S.np, S.nx, S.nu, S.ny, NT, NSUBr);
    P.image.assign(L.total, 0.0);
    // [x0 | a | b | c] over [y0 | dy | ey | fy]: through Layout::lin (x rows 0 .. nx-1 of the SHAPE, y rows after them)
    auto lin = [&](int col, int row) -> double & { return P.image[L.lin(col, row, S.nx, S.nu)]; };
    for (int i = 0; i < d.nx; ++i) {
        lin(0, i) = m.x0[i];
        for (int j = 0; j < d.nx; ++j) lin(1 + j, i) = m.a[(size_t)j * d.nx + i];
        for (int k = 0; k < d.nu; ++k) lin(1 + S.nx + k, i) = m.b[(size_t)k * d.nx + i];
    }
    for (int i = 0; i < d.ny; ++i) {
        lin(0, S.nx + i) = m.y0[i];
        for (int j = 0; j < d.nx; ++j) lin(1 + j, S.nx + i) = m.dy[(size_t)j * d.ny + i];
        for (int k = 0; k < d.nu; ++k) lin(1 + S.nx + k, S.nx + i) = m.ey[(size_t)k * d.ny + i];
    }
    P.rowc.assign((size_t)NSUBr * ROWC * GROUP, 0.0);
    P.rowi.assign((size_t)NSUBr * ROWI * GROUP, 0);
    P.init_state.assign((size_t)S.nx + (size_t)NSUBr * (S.np + S.nn), 0.0);
    P.nterms = 2; P.has_bjt = 0; P.rare_kinds = 0;
    // actual z offsets of the sub-problems inside the reference's z vector
    std::vector<int> zoff(m.subs.size() + 1, 0);
    for (size_t k = 0; k < m.subs.size(); ++k) zoff[k + 1] = zoff[k] + m.subs[k].nn;
    for (size_t k = 0; k < m.subs.size(); ++k) {
        const HostSub *s = &m.subs[k];
        const int base = L.sub0 + (int)k * L.sub_stride;
        // c and fy: this sub-problem's z columns go to padded columns k*NN ...
        for (int j = 0; j < s->nn; ++j) {
            for (int i = 0; i < d.nx; ++i) lin(1 + S.nx + S.nu + (int)k * S.nn + j, i) = m.c[(size_t)(zoff[k] + j) * d.nx + i];
            for (int i = 0; i < d.ny; ++i) lin(1 + S.nx + S.nu + (int)k * S.nn + j, S.nx + i) = m.fy[(size_t)(zoff[k] + j) * d.ny + i];
        }
        for (int i = 0; i < s->np; ++i) {      // [dq | eq] through Layout::pq
            for (int j = 0; j < d.nx; ++j) P.image[base + L.pq(j, i, S.np, S.nx)] = s->dq[(size_t)j * s->np + i];
            for (int kk = 0; kk < d.nu; ++kk) P.image[base + L.pq(S.nx + kk, i, S.np, S.nx)] = s->eq[(size_t)kk * s->np + i];
        }
        for (size_t kp = 0; kp < k; ++kp)  // fqprev: columns of the earlier sub-problems
            for (int j = 0; j < m.subs[kp].nn; ++j)
                for (int i = 0; i < s->np; ++i)
                    P.image[base + L.fqprev + ((size_t)kp * S.nn + j) * S.np + i] = s->fqprev[(size_t)(zoff[kp] + j) * s->np + i];
        for (int i = 0; i < s->nn; ++i) P.init_state[(size_t)S.nx + (size_t)NSUBr * S.np + k * S.nn + i] = s->init_z[i];

        double *rowc = &P.rowc[k * ROWC * GROUP];
        int *rowi = &P.rowi[k * ROWI * GROUP];
        std::vector<int> pos_of(s->nn);
        for (int i = 0; i < s->nn; ++i) pos_of[i] = i;
        if (!s->row_order.empty()) {
            if ((int)s->row_order.size() != s->nn) { err = "row_order has the wrong length"; return false; }
            std::vector<int> seen(s->nn, 0);
            for (int pos = 0; pos < s->nn; ++pos) {
                int r = s->row_order[pos];
                if (r < 0 || r >= s->nn || seen[r]) { err = "row_order is not a permutation"; return false; }
                seen[r] = 1;
                pos_of[r] = pos;
            }
        }
        for (size_t e = 0; e < s->kind.size(); ++e) {
            int kind = s->kind[e], knq, knn;
            kind_shape(kind, knq, knn);
            for (int er = 0; er < knn; ++er) {
                int row = s->roff[e] + er;
                if (row >= s->nn) { err = "element table row out of range"; return false; }
                double rc[ROWC];
                int ri[ROWI];
                if (!describe_element(kind, &s->par[e * MAX_ELEM_PAR], er, s->qoff[e], rc, ri, P, err)) return false;
                for (int c = 0; c < ROWC; ++c) rowc[(size_t)c * GROUP + pos_of[row]] = rc[c];
                for (int w = 0; w < ROWI; ++w) rowi[(size_t)w * GROUP + pos_of[row]] = ri[w];
            }
        }
        // shape padding: extra unknowns z_pad with the trivial equation q_pad = z_pad = 0
        std::vector<double> fqp((size_t)S.nq * S.nn, 0.0), pexpp((size_t)S.nq * S.np, 0.0), q0p(S.nq, 0.0);
        put(fqp, 0, S.nq, s->fq, s->nq, s->nn);
        put(pexpp, 0, S.nq, s->pexp, s->nq, s->np);
        put(q0p, 0, S.nq, s->q0, s->nq, 1);
        for (int r = s->nn; r < S.nn; ++r) {
            int qrow = s->nq + (r - s->nn);
            fqp[(size_t)r * S.nq + qrow] = 1.0;
            rowi[0 * GROUP + r] = RK_PAD;
            rowc[(size_t)UR_G0 * GROUP + r] = 1.0;
            for (int t = 0; t < 4; ++t) rowi[(3 + t) * GROUP + r] = qrow;
        }
        // condensed shapes: row 15 (never a residual row, static_assert in Shape) holds the constants the lanes of
        // the LINEAR rows evaluate in the Newton loop -- res = e0, Jq = (1, 0, 0): see acme_kernel.h
        if (S.nl > 0) rowc[(size_t)UR_G0 * GROUP + (GROUP - 1)] = 1.0;
        // row-gathered copies of fq / pexp / q0 (see Layout): slot `pos` = the lane position of
        // the residual row, term t = its t-th Jq non-zero
        for (int pos = 0; pos < S.nn; ++pos)
            for (int t = 0; t < NT; ++t) {
                int tc = rowi[(3 + t) * GROUP + pos];
                for (int j = 0; j < S.nn; ++j)
                    P.image[base + L.fqr + L.gat(t, j, pos, S.nn)] = fqp[(size_t)j * S.nq + tc];
                for (int j = 0; j < S.np; ++j)
                    P.image[base + L.pexpr + L.gat(t, j, pos, S.np)] = pexpp[(size_t)j * S.nq + tc];
                P.image[base + L.q0i(t, pos)] = q0p[tc];
                if (L.pairs && (S.np % 2 == 1))     // the pad column of the last pexp pair: q0 comes with the pair read
                    P.image[base + L.pexpr + L.gat(t, S.np, pos, S.np)] = q0p[tc];
            }
    }
    // the lane-per-instance kernel's constant block (see LaneLayout): regrouped copies of the above
    P.lanec.clear();
    if (m.subs.size() == 1 && !S.rare) {
        const LaneLayout ll = make_lane_layout(S.nn, S.np, S.nx, S.nu, S.ny);
        P.lanec.assign(ll.total, 0.0);
        const int base = L.sub0;
        for (int r = 0; r < S.nn; ++r) {
            double *row = &P.lanec[(size_t)r * ll.row];
            for (int t = 0; t < 3; ++t) {
                row[ll.q0 + t] = P.image[base + L.q0i(t, r)];
                for (int j = 0; j < S.np; ++j) row[ll.pexp + t * S.np + j] = P.image[base + L.pexpr + L.gat(t, j, r, S.np)];
                for (int j = 0; j < S.nn; ++j) row[ll.fq + t * S.nn + j] = P.image[base + L.fqr + L.gat(t, j, r, S.nn)];
            }
            for (int c = UR_SA; c <= UR_W1; ++c) row[ll.ur + c - UR_SA] = P.rowc[(size_t)c * GROUP + r];
            row[ll.kind] = (double)P.rowi[0 * GROUP + r];
        }
        for (int i = 0; i < S.np; ++i) {
            double *row = &P.lanec[ll.p0 + (size_t)i * ll.pstr];
            for (int j = 0; j < S.nx + S.nu; ++j) row[j] = P.image[base + L.pq(j, i, S.np, S.nx)];
        }
        for (int which = 0; which < 2; ++which) {       // 0: y rows, 1: x rows
            const int rows = which ? S.nx : S.ny, r0 = which ? 0 : S.nx;   // rows of [a; dy] etc.: x rows first, then y rows
            for (int i = 0; i < rows; ++i) {
                double *row = &P.lanec[(which ? ll.x0 : ll.y0) + (size_t)i * ll.xstr];
                for (int c = 0; c < 1 + S.nx + S.nu + S.nn; ++c) row[c] = lin(c, r0 + i);
            }
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// The generic kernel's model (acme_generic.h): any dimensions, any number of sub-problems.  Matrices column-major as
// the caller handed them over; one row descriptor per residual row, 16 rows to a block of the row tables.
// ---------------------------------------------------------------------------------------------------------------
struct PackedGeneric {
    GenHeader H{};
    std::vector<double> image, rowc, init_state;
    std::vector<int> rowi;
};
// like: the header of the batch this model is to join as ONE instance's model (acme_batch_set_matrices) -- its sparse forms
// take the batch's entries per row, so that every instance's image has the batch's layout; a model with a fuller row than
// that comes back with GenHeader::ell == 0 (the caller then runs the whole batch on the dense matrices).
inline bool pack_generic(const HostModel &m, PackedGeneric &G, std::string &err, const GenHeader *like = nullptr) {
    GenHeader &H = G.H;
    H = GenHeader{};
    if ((int)m.subs.size() > GEN_MAX_SUB) {
        err = "more than 64 nonlinear sub-problems";
        return false;
    }
    H.nx = m.nx; H.nu = m.nu; H.ny = m.ny; H.nsub = (int)m.subs.size();
    int o = 0, rows = 0;
    auto take = [&](int n) { int at = o; o += n; return at; };
    int nnt = 0, npt = 0;
    for (const HostSub &s : m.subs) { nnt += s.nn; npt += s.np; }
    H.nnt = nnt; H.npt = npt;
    H.o_a = take(m.nx * m.nx); H.o_b = take(m.nx * m.nu); H.o_c = take(m.nx * nnt); H.o_x0 = take(m.nx);
    H.o_dy = take(m.ny * m.nx); H.o_ey = take(m.ny * m.nu); H.o_fy = take(m.ny * nnt); H.o_y0 = take(m.ny);
    int zoff = 0, poff = 0, coff = 0;
    for (int k = 0; k < H.nsub; ++k) {
        const HostSub &s = m.subs[k];
        GenSub &g = H.sub[k];
        g.nn = s.nn; g.nq = s.nq; g.np = s.np; g.zoff = zoff; g.poff = poff;
        g.o_pexp = take(s.nq * s.np); g.o_dq = take(s.np * m.nx); g.o_eq = take(s.np * m.nu);
        g.o_fqprev = take(s.np * nnt); g.o_fq = take(s.nq * s.nn); g.o_q0 = take(s.nq);
        g.row0 = rows;
        g.c_off = coff;
        rows += s.nn;
        zoff += s.nn; poff += s.np;
        coff += (s.np + s.nn) * CACHE + 2;
        if (s.nn > H.nnmax) H.nnmax = s.nn;
        if (s.nq > H.nqmax) H.nqmax = s.nq;
        if (s.np > H.npmax) H.npmax = s.np;
    }
    H.image_total = o > 0 ? o : 1;
    H.cache_total = coff > 0 ? coff : 1;
    H.state_total = m.nx + npt + nnt;
    G.image.assign(H.image_total, 0.0);
    auto putm = [&](int at, const std::vector<double> &v) { for (size_t i = 0; i < v.size(); ++i) G.image[at + i] = v[i]; };
    putm(H.o_a, m.a); putm(H.o_b, m.b); putm(H.o_c, m.c); putm(H.o_x0, m.x0);
    putm(H.o_dy, m.dy); putm(H.o_ey, m.ey); putm(H.o_fy, m.fy); putm(H.o_y0, m.y0);
    const int blocks = (rows + GROUP - 1) / GROUP;
    G.rowc.assign((size_t)(blocks > 0 ? blocks : 1) * ROWC * GROUP, 0.0);
    G.rowi.assign((size_t)(blocks > 0 ? blocks : 1) * ROWI * GROUP, 0);
    G.init_state.assign(H.state_total > 0 ? H.state_total : 1, 0.0);
    Packed dummy;       // (describe_element notes has_bjt / rare kinds there)
    for (int k = 0; k < H.nsub; ++k) {
        const HostSub &s = m.subs[k];
        const GenSub &g = H.sub[k];
        putm(g.o_pexp, s.pexp); putm(g.o_dq, s.dq); putm(g.o_eq, s.eq); putm(g.o_fqprev, s.fqprev);
        putm(g.o_fq, s.fq); putm(g.o_q0, s.q0);
        for (int i = 0; i < s.nn; ++i) G.init_state[m.nx + npt + g.zoff + i] = s.init_z[i];
        std::vector<char> have(s.nn, 0);
        for (size_t e = 0; e < s.kind.size(); ++e) {
            int knq, knn;
            kind_shape(s.kind[e], knq, knn);
            for (int er = 0; er < knn; ++er) {
                const int row = s.roff[e] + er;
                if (row < 0 || row >= s.nn || s.qoff[e] < 0 || s.qoff[e] + knq > s.nq) { err = "element table out of range"; return false; }
                double rc[ROWC];
                int ri[ROWI];
                if (!describe_element(s.kind[e], &s.par[e * MAX_ELEM_PAR], er, s.qoff[e], rc, ri, dummy, err)) return false;
                const int R = g.row0 + row, blk = R / GROUP, ln = R % GROUP;
                for (int c = 0; c < ROWC; ++c) G.rowc[((size_t)blk * ROWC + c) * GROUP + ln] = rc[c];
                for (int w = 0; w < ROWI; ++w) G.rowi[((size_t)blk * ROWI + w) * GROUP + ln] = ri[w];
                have[row] = 1;
            }
        }
        for (int r = 0; r < s.nn; ++r)
            if (!have[r]) { err = "a residual row belongs to no element"; return false; }
    }
    H.has_bjt = dummy.has_bjt;
    // ---- the sparse forms of the matrices (the mid-size kernel, acme_coop.h: one sub-problem or none), behind the dense ones ----
    H.ell = 0;
    H.o_ell = H.o_jtab = H.image_total;
    if (H.nsub <= 1) {
        const int base = (H.image_total + 1) & ~1;
        std::vector<double> tail;
        bool fits = true;
        auto at = [&]() { return base + (int)tail.size(); };
        auto build = [&](GenEll &E, int nrows, int ncols, auto get, const GenEll *lk) {
            int k = 0;
            for (int r = 0; r < nrows; ++r) {
                int cnt = 0;
                for (int c = 0; c < ncols; ++c) cnt += get(r, c) != 0.0;
                if (cnt > k) k = cnt;
            }
            if (lk) { if (k > lk->k) fits = false; k = lk->k; }
            E.rows = nrows; E.k = k;
            E.o_val = at(); tail.resize(tail.size() + (size_t)nrows * k, 0.0);
            E.o_col = at(); tail.resize(tail.size() + (size_t)nrows * k, 0.0);
            for (int r = 0; r < nrows; ++r) {
                int e = 0;
                for (int c = 0; c < ncols && e < k; ++c) {
                    const double v = get(r, c);
                    if (v != 0.0) {
                        tail[E.o_val - base + (size_t)e * nrows + r] = v;
                        tail[E.o_col - base + (size_t)e * nrows + r] = (double)c;
                        ++e;
                    }
                }
            }
        };
        const int nxy = m.nx + m.ny;
        build(H.e_ax, nxy, m.nx, [&](int r, int c) { return r < m.nx ? m.a[(size_t)c * m.nx + r] : m.dy[(size_t)c * m.ny + (r - m.nx)]; }, like ? &like->e_ax : nullptr);
        build(H.e_bu, nxy, m.nu, [&](int r, int c) { return r < m.nx ? m.b[(size_t)c * m.nx + r] : m.ey[(size_t)c * m.ny + (r - m.nx)]; }, like ? &like->e_bu : nullptr);
        build(H.e_cz, nxy, nnt, [&](int r, int c) { return r < m.nx ? m.c[(size_t)c * m.nx + r] : m.fy[(size_t)c * m.ny + (r - m.nx)]; }, like ? &like->e_cz : nullptr);
        H.o_xy0 = at();
        tail.insert(tail.end(), m.x0.begin(), m.x0.end());
        tail.insert(tail.end(), m.y0.begin(), m.y0.end());
        if (H.nsub == 1) {
            const HostSub &s = m.subs[0];
            GenSub &g = H.sub[0];
            const GenSub *lg = like ? &like->sub[0] : nullptr;
            build(g.e_fq, s.nq, s.nn, [&](int r, int c) { return s.fq[(size_t)c * s.nq + r]; }, lg ? &lg->e_fq : nullptr);
            build(g.e_pexp, s.nq, s.np, [&](int r, int c) { return s.pexp[(size_t)c * s.nq + r]; }, lg ? &lg->e_pexp : nullptr);
            build(g.e_dq, s.np, m.nx, [&](int r, int c) { return s.dq[(size_t)c * s.np + r]; }, lg ? &lg->e_dq : nullptr);
            build(g.e_eq, s.np, m.nu, [&](int r, int c) { return s.eq[(size_t)c * s.np + r]; }, lg ? &lg->e_eq : nullptr);
            g.o_q0s = at();
            tail.insert(tail.end(), s.q0.begin(), s.q0.end());
            // the rows of J = Jq fq and Jp = Jq pexp: a residual row's four Jq terms sit at the q rows its table entry names
            auto tc_of = [&](int r, int t) {
                const int R = g.row0 + r;
                return G.rowi[((size_t)(R / GROUP) * ROWI + 3 + t) * GROUP + R % GROUP];
            };
            auto rows_of = [&](const std::vector<double> &mat, int ncols, int pad_col, int like_k, int &k, int &o_col, int &o_coef) {
                k = 0;
                for (int r = 0; r < s.nn; ++r) {
                    int cnt = 0;
                    for (int c = 0; c < ncols; ++c) {
                        bool any = false;
                        for (int t = 0; t < 4; ++t) any = any || mat[(size_t)c * s.nq + tc_of(r, t)] != 0.0;
                        cnt += any;
                    }
                    if (cnt > k) k = cnt;
                }
                if (like_k >= 0) { if (k > like_k) fits = false; k = like_k; }
                o_col = at(); tail.resize(tail.size() + (size_t)s.nn * k, (double)pad_col);
                o_coef = at(); tail.resize(tail.size() + (size_t)s.nn * k * 4, 0.0);
                for (int r = 0; r < s.nn; ++r) {
                    int e = 0;
                    for (int c = 0; c < ncols && e < k; ++c) {
                        bool any = false;
                        for (int t = 0; t < 4; ++t) any = any || mat[(size_t)c * s.nq + tc_of(r, t)] != 0.0;
                        if (!any) continue;
                        tail[o_col - base + (size_t)e * s.nn + r] = (double)c;
                        for (int t = 0; t < 4; ++t) tail[o_coef - base + ((size_t)e * 4 + t) * s.nn + r] = mat[(size_t)c * s.nq + tc_of(r, t)];
                        ++e;
                    }
                }
            };
            H.o_jtab = at();
            rows_of(s.fq, s.nn, s.nn + 1, lg ? lg->kj : -1, g.kj, g.o_jcol, g.o_jcoef);
            rows_of(s.pexp, s.np, s.np, lg ? lg->kp : -1, g.kp, g.o_pcol, g.o_pcoef);
        }
        if (H.nsub != 1) H.o_jtab = at();
        H.o_ell = base;
        H.ell = fits ? 1 : 0;
        G.image.resize((size_t)base, 0.0);
        G.image.insert(G.image.end(), tail.begin(), tail.end());
        H.image_total = (int)G.image.size();
    }
    // workspace
    // (factor matrices: room for the cooperative kernel's row pitch -- at least 3 columns of slack for its 4-wide updates,
    // and a pitch of 2 mod 4 doubles so that the 16 rows a DPP row of lanes touches in one LDS access fall into 16 distinct
    // bank groups, acme_coop.h)
    H.ldf = H.nnmax + 3;
    while (H.ldf % 4 != 2) ++H.ldf;
    int w = 0;
    auto wt = [&](int n) { int at = w; w += n; return at; };
    auto wt16 = [&](int n) { w = (w + 1) & ~1; return wt(n); };      // 16-byte aligned: the factor matrices (ds_read_b128 rows)
    H.w_x = wt(m.nx); H.w_xn = wt(m.nx); H.w_z = wt(nnt);
    for (int k = 0; k < H.nsub; ++k) {
        GenSub &g = H.sub[k];
        g.w_lp = wt(g.np); g.w_lz = wt(g.nn); g.w_ljp = wt(g.nn * g.np); g.w_llu = wt16(g.nn * H.ldf); g.w_lpiv = wt(g.nn);
    }
    H.w_p = wt(H.npmax); H.w_pa = wt(H.npmax); H.w_sp = wt(H.npmax); H.w_zz = wt(H.nnmax); H.w_res = wt(H.nnmax);
    H.w_dz = wt(H.nnmax); H.w_lu = wt16(H.nnmax * H.ldf); H.w_piv = wt(H.nnmax); H.w_jp = wt(H.nnmax * H.npmax);
    H.w_q = wt(H.nqmax); H.w_pf = wt(H.nqmax); H.w_tv = wt(4 * H.nnmax); H.w_tmp = wt(H.nnmax); H.w_u = wt(m.nu);
    H.ws_total = w > 0 ? w : 1;
    return true;
}

}  // namespace acme
