/* acme_ref.c -- CPU ORACLE (test infrastructure, NOT product code).  See acme_ref.h.
 *
 * Scalar double-precision restatement of the reference's run!/step!/solver stack,
 * operation for operation (same op order in step!, same LU pivot rule, same Newton loop
 * order "evaluate -> finite check -> LU -> converged? -> step", same homotopy schedule).
 * Compile WITHOUT -ffast-math and with -ffp-contract=off so that no FMA contraction
 * happens (Julia does not contract a*b+c either).
 *
 * Only deviation from the letter of the reference: CachingSolver's k-d tree
 * (src/kdtree.jl) is replaced by an exhaustive nearest-neighbour scan.  The tree search
 * is exact (best-bin-first with full backtracking), so both return the same stored
 * point; only exact distance ties could resolve differently.
 */
#include "acme_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define MAXPAR ACME_REF_MAX_PAR

/* ------------------------------------------------------------------------------------ */
/* element nonlinearities (src/elements.jl)                                              */
/* ------------------------------------------------------------------------------------ */
static double evalpoly(double x, const double *c, int n) {
    /* Base.evalpoly: Horner */
    double acc = c[n - 1];
    for (int k = n - 2; k >= 0; --k) acc = acc * x + c[k];
    return acc;
}

static double sgn(double v) { return (v > 0) - (v < 0); }

void acme_ref_eval_element(int kind, const double *par, const double *q, double *res,
                           double *Jq) {
    switch (kind) {
    case ACME_REF_KIND_DIODE: { /* src/elements.jl:238-244 */
        double is = par[0], eta = par[1];
        double v = q[0], i = q[1];
        double ex = exp(v * (1 / (25e-3 * eta)));
        res[0] = is * (ex - 1) - i;
        Jq[0] = is / (25e-3 * eta) * ex;
        Jq[1] = -1;
        break;
    }
    case ACME_REF_KIND_POT: { /* src/elements.jl:25-30 */
        double r = par[0];
        double v1 = q[0], v2 = q[1], i1 = q[2], i2 = q[3], pos = q[4];
        res[0] = v1 - r * pos * i1;
        res[1] = v2 - r * (1 - pos) * i2;
        double J[10] = {1, 0, -r * pos, 0, -r * i1, 0, 1, 0, -r * (1 - pos), -r * i2};
        memcpy(Jq, J, sizeof J);
        break;
    }
    case ACME_REF_KIND_BJT: { /* src/elements.jl:323-401 */
        double ise = par[0], isc = par[1], etae = par[2], etac = par[3], bf = par[4],
               br = par[5], ile = par[6], ilc = par[7], etael = par[8], etacl = par[9],
               vaf = par[10], var = par[11], ikf = par[12], ikr = par[13];
        double vE = q[0], vC = q[1], iE = q[2], iC = q[3];
        double expE = exp(vE * (1 / (25e-3 * etae)));
        double expC = exp(vC * (1 / (25e-3 * etac)));
        double i_f = (bf / (1 + bf) * ise) * (expE - 1);
        double i_r = (br / (1 + br) * isc) * (expC - 1);
        double di_f1 = (bf / (1 + bf) * ise / (25e-3 * etae)) * expE;
        double di_r2 = (br / (1 + br) * isc / (25e-3 * etac)) * expC;
        double i_cc, di_cc1, di_cc2;
        int early = !(var == INFINITY && vaf == INFINITY);
        int knee = !(ikf == INFINITY && ikr == INFINITY);
        if (!early && !knee) { /* :331-334 */
            i_cc = i_f - i_r;
            di_cc1 = di_f1;
            di_cc2 = -di_r2;
        } else if (early && !knee) { /* :335-343 */
            double q1i = 1 - vE * (1 / var) - vC * (1 / vaf);
            i_cc = q1i * (i_f - i_r);
            double dq1 = (-1 / var), dq2 = (-1 / vaf);
            di_cc1 = dq1 * (i_f - i_r) + q1i * di_f1;
            di_cc2 = dq2 * (i_f - i_r) - q1i * di_r2;
        } else if (!early && knee) { /* :344-356 */
            double q2 = i_f * (1 / ikf) + i_r * (1 / ikr);
            double qden = 1 + sqrt(1 + 4 * q2);
            double qfact = 2 / qden;
            i_cc = qfact * (i_f - i_r);
            double dq21 = di_f1 * (1 / ikf);
            double dq22 = di_r2 * (1 / ikr);
            double dqfact1 = -4 * dq21 / (qden - 1) / (qden * qden);
            double dqfact2 = -4 * dq22 / (qden - 1) / (qden * qden);
            di_cc1 = dqfact1 * (i_f - i_r) + qfact * di_f1;
            di_cc2 = dqfact2 * (i_f - i_r) - qfact * di_r2;
        } else { /* :357-373 */
            double q1i = 1 - vE * (1 / var) - vC * (1 / vaf);
            double q2 = i_f * (1 / ikf) + i_r * (1 / ikr);
            double qden = 1 + sqrt(1 + 4 * q2);
            double qfact = 2 * q1i / qden;
            i_cc = qfact * (i_f - i_r);
            double dq11 = -1 / var, dq12 = -1 / vaf;
            double dq21 = di_f1 * (1 / ikf);
            double dq22 = di_r2 * (1 / ikr);
            double dqfact1 = (2 * dq11 * qden - q1i * 4 * dq21 / (qden - 1)) / (qden * qden);
            double dqfact2 = (2 * dq12 * qden - q1i * 4 * dq22 / (qden - 1)) / (qden * qden);
            di_cc1 = dqfact1 * (i_f - i_r) + qfact * di_f1;
            di_cc2 = dqfact2 * (i_f - i_r) - qfact * di_r2;
        }
        double iBE = (1 / bf) * i_f;
        double diBE1 = (1 / bf) * di_f1;
        if (ile != 0) { /* :377-385 (derivative uses etae as in the reference) */
            double expEl = (etael != etae) ? exp(vE * (1 / (25e-3 * etael))) : expE;
            iBE += ile * (expEl - 1);
            diBE1 += (ile / (25e-3 * etae)) * expEl;
        }
        double iBC = (1 / br) * i_r;
        double diBC2 = (1 / br) * di_r2;
        if (ilc != 0) { /* :388-396 */
            double expCl = (etacl != etac) ? exp(vC * (1 / (25e-3 * etacl))) : expC;
            iBC += ilc * (expCl - 1);
            diBC2 += (ilc / (25e-3 * etac)) * expCl;
        }
        res[0] = i_cc + iBE - iE;
        res[1] = -i_cc + iBC - iC;
        double J[8] = {di_cc1 + diBE1, di_cc2, -1.0, 0.0, -di_cc1, -di_cc2 + diBC2, 0.0, -1.0};
        memcpy(Jq, J, sizeof J);
        break;
    }
    case ACME_REF_KIND_MOSFET: { /* src/elements.jl:453-479 */
        double pol = par[0], lam = par[1];
        int nvt = (int)par[2];
        const double *vt = par + 3;
        int na = (int)par[7];
        const double *al = par + 8;
        double dvt[4] = {0, 0, 0, 0}, dal[4] = {0, 0, 0, 0};
        for (int k = 1; k < nvt; ++k) dvt[k - 1] = vt[k] * k;
        for (int k = 1; k < na; ++k) dal[k - 1] = al[k] * k;
        double vgs = q[0], vds = q[1], id = q[2];
        double a_ = evalpoly(pol * vgs, al, na);
        double da = (na > 1) ? evalpoly(pol * vgs, dal, na - 1) : 0;
        double vt_ = evalpoly(pol * vgs, vt, nvt);
        double dvt_ = (nvt > 1) ? evalpoly(pol * vgs, dvt, nvt - 1) : 0;
        double lam_ = vds >= 0 ? lam : 0;
        if (vgs <= vt_) {
            res[0] = -id;
            Jq[0] = 0;
            Jq[1] = 0;
            Jq[2] = -1;
        } else if (vds <= vgs - vt_) {
            res[0] = a_ * (vgs - vt_ - 0.5 * vds) * vds * (1 + lam_ * vds) - id;
            Jq[0] = a_ * (1 - dvt_) * vds * (1 + lam_ * vds) +
                    da * (vgs - vt_ - 0.5 * vds) * vds * (1 + lam_ * vds);
            Jq[1] = a_ * (vgs - vt_ + vds * (2 * lam_ * (vgs - vt_ - 0.75 * vds) - 1));
            Jq[2] = -1;
        } else {
            double d = vgs - vt_;
            res[0] = (a_ / 2) * (d * d) * (1 + lam_ * vds) - id;
            Jq[0] = a_ * d * (1 - dvt_) * (1 + lam_ * vds) + da / 2 * (d * d) * (1 + lam_ * vds);
            Jq[1] = lam_ * a_ / 2 * (d * d);
            Jq[2] = -1;
        }
        break;
    }
    case ACME_REF_KIND_MACAK: { /* src/elements.jl:540-546 */
        double gain = par[0], scale = par[1];
        double vi = q[0], vo = q[1];
        double vs = vi * (gain / scale);
        double ch = cosh(vs);
        res[0] = tanh(vs) * scale - vo;
        Jq[0] = gain / (ch * ch);
        Jq[1] = -1;
        break;
    }
    case ACME_REF_KIND_JA: { /* src/elements.jl:107-129 */
        double Ms = par[0], a = par[1], alpha = par[2], c = par[3], k = par[4];
        double q1 = q[0], q2 = q[1], q3 = q[2], q4 = q[3];
        double coth = 1 / tanh(q1);
        double aq1 = fabs(q1);
        double L = aq1 < 1e-4 ? q1 / 3 : coth - 1 / q1;
        double Ld = aq1 < 1e-4 ? 1.0 / 3 : 1 / (q1 * q1) - coth * coth + 1;
        double Ld2 = aq1 < 1e-3 ? -2.0 / 15 * q1
                                : 2 * coth * (coth * coth - 1) - 2 / (q1 * q1 * q1);
        double delta = q3 > 0 ? 1.0 : -1.0;
        double Man = Ms * L;
        double dM = sgn(q3) == sgn(Man - q2) ? 1.0 : 0.0;
        double den = delta * (k * (1 - c)) - alpha * (Man - q2);
        res[0] = (1e-4 / Ms) * ((1 - c) * dM * (Man - q2) / den * q3 +
                                (c * Ms / a) * (q3 + alpha * q4) * Ld - q4);
        Jq[0] = (1e-4 / Ms) * (((1 - c) * (1 - c) * k * Ms) * dM * Ld * delta / (den * den) * q3 +
                               (c * Ms / a) * (q3 + alpha * q4) * Ld2);
        Jq[1] = (1e-4 / Ms) * -((1 - c) * (1 - c)) * k * dM * delta / (den * den) * q3;
        Jq[2] = (1e-4 / Ms) * ((1 - c) * dM * (Man - q2) / den + (c * Ms / a) * Ld);
        Jq[3] = (1e-4 / Ms) * ((c * Ms / a * alpha) * Ld - 1);
        break;
    }
    default:
        break;
    }
}

/* ------------------------------------------------------------------------------------ */
/* model                                                                                 */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int kind, nq, nn, qoff, roff;
    double par[MAXPAR];
} ref_elem;

typedef struct {
    int nn, nq, np;
    double *pexp, *dq, *eq, *fqprev, *fq, *q0, *init_z;
    int n_elems;
    ref_elem *elems;
} ref_sub;

struct acme_ref_model {
    int nx, nu, ny, nn_total, nsub;
    double *a, *b, *c, *x0, *dy, *ey, *fy, *y0;
    ref_sub *subs;
};

static double *dupd(const double *src, size_t n) {
    double *d = (double *)calloc(n ? n : 1, sizeof(double));
    if (src && n) memcpy(d, src, n * sizeof(double));
    return d;
}

static void kind_shape(int kind, int *nq, int *nn) {
    switch (kind) {
    case ACME_REF_KIND_DIODE: *nq = 2; *nn = 1; break;
    case ACME_REF_KIND_BJT: *nq = 4; *nn = 2; break;
    case ACME_REF_KIND_POT: *nq = 5; *nn = 2; break;
    case ACME_REF_KIND_MOSFET: *nq = 3; *nn = 1; break;
    case ACME_REF_KIND_MACAK: *nq = 2; *nn = 1; break;
    case ACME_REF_KIND_JA: *nq = 4; *nn = 1; break;
    default: *nq = 0; *nn = 0; break;
    }
}

acme_ref_model *acme_ref_model_create(int nx, int nu, int ny, int nn_total, const double *a,
                                      const double *b, const double *c, const double *x0,
                                      const double *dy, const double *ey, const double *fy,
                                      const double *y0) {
    acme_ref_model *m = (acme_ref_model *)calloc(1, sizeof *m);
    m->nx = nx; m->nu = nu; m->ny = ny; m->nn_total = nn_total;
    m->a = dupd(a, (size_t)nx * nx);
    m->b = dupd(b, (size_t)nx * nu);
    m->c = dupd(c, (size_t)nx * nn_total);
    m->x0 = dupd(x0, nx);
    m->dy = dupd(dy, (size_t)ny * nx);
    m->ey = dupd(ey, (size_t)ny * nu);
    m->fy = dupd(fy, (size_t)ny * nn_total);
    m->y0 = dupd(y0, ny);
    return m;
}

int acme_ref_model_add_subproblem(acme_ref_model *m, int nn, int nq, int np, const double *pexp,
                                  const double *dq, const double *eq, const double *fqprev,
                                  const double *fq, const double *q0, const double *init_z,
                                  int n_elems, const int *elem_kind, const int *elem_qoff,
                                  const int *elem_roff, const double *elem_par) {
    m->subs = (ref_sub *)realloc(m->subs, (size_t)(m->nsub + 1) * sizeof(ref_sub));
    ref_sub *s = &m->subs[m->nsub];
    memset(s, 0, sizeof *s);
    s->nn = nn; s->nq = nq; s->np = np;
    s->pexp = dupd(pexp, (size_t)nq * np);
    s->dq = dupd(dq, (size_t)np * m->nx);
    s->eq = dupd(eq, (size_t)np * m->nu);
    s->fqprev = dupd(fqprev, (size_t)np * m->nn_total);
    s->fq = dupd(fq, (size_t)nq * nn);
    s->q0 = dupd(q0, nq);
    s->init_z = dupd(init_z, nn);
    s->n_elems = n_elems;
    s->elems = (ref_elem *)calloc(n_elems ? n_elems : 1, sizeof(ref_elem));
    for (int e = 0; e < n_elems; ++e) {
        s->elems[e].kind = elem_kind[e];
        kind_shape(elem_kind[e], &s->elems[e].nq, &s->elems[e].nn);
        s->elems[e].qoff = elem_qoff[e];
        s->elems[e].roff = elem_roff[e];
        memcpy(s->elems[e].par, elem_par + (size_t)e * MAXPAR, MAXPAR * sizeof(double));
    }
    return m->nsub++;
}

void acme_ref_model_destroy(acme_ref_model *m) {
    if (!m) return;
    for (int i = 0; i < m->nsub; ++i) {
        ref_sub *s = &m->subs[i];
        free(s->pexp); free(s->dq); free(s->eq); free(s->fqprev); free(s->fq); free(s->q0);
        free(s->init_z); free(s->elems);
    }
    free(m->subs);
    free(m->a); free(m->b); free(m->c); free(m->x0); free(m->dy); free(m->ey); free(m->fy);
    free(m->y0);
    free(m);
}

/* ------------------------------------------------------------------------------------ */
/* LinearSolver (src/solvers.jl:38-132)                                                  */
/* ------------------------------------------------------------------------------------ */
static long long g_lu_swaps, g_lu_count; /* diagnostics only */

int acme_ref_lu_factor(int n, double *f, int *ipiv) {
    /* setlhs! (:46-96): partial pivoting, first strict maximum, full-row interchange,
     * reciprocal stored on the diagonal, false on an exactly-zero pivot */
    ++g_lu_count;
    for (int k = 0; k < n; ++k) {
        int kp = k;
        double amax = 0.0;
        for (int i = k; i < n; ++i) {
            double absi = fabs(f[i + (size_t)k * n]);
            if (absi > amax) {
                kp = i;
                amax = absi;
            }
        }
        ipiv[k] = kp;
        if (f[kp + (size_t)k * n] != 0.0) {
            if (k != kp) {
                ++g_lu_swaps;
                for (int i = 0; i < n; ++i) {
                    double tmp = f[k + (size_t)i * n];
                    f[k + (size_t)i * n] = f[kp + (size_t)i * n];
                    f[kp + (size_t)i * n] = tmp;
                }
            }
            double inv = 1.0 / f[k + (size_t)k * n];
            f[k + (size_t)k * n] = inv;
            for (int i = k + 1; i < n; ++i) f[i + (size_t)k * n] *= inv;
        } else {
            return 0;
        }
        for (int j = k + 1; j < n; ++j)
            for (int i = k + 1; i < n; ++i)
                f[i + (size_t)j * n] -= f[i + (size_t)k * n] * f[k + (size_t)j * n];
    }
    return 1;
}

void acme_ref_lu_solve(int n, const double *f, const int *ipiv, double *x) {
    /* solve! (:98-132) */
    for (int i = 0; i < n; ++i) {
        double t = x[i];
        x[i] = x[ipiv[i]];
        x[ipiv[i]] = t;
    }
    for (int j = 0; j < n; ++j) {
        double xj = x[j];
        for (int i = j + 1; i < n; ++i) x[i] -= f[i + (size_t)j * n] * xj;
    }
    for (int j = n - 1; j >= 0; --j) {
        double xj = x[j] = f[j + (size_t)j * n] * x[j];
        for (int i = 0; i < j; ++i) x[i] -= f[i + (size_t)j * n] * xj;
    }
}

/* ------------------------------------------------------------------------------------ */
/* ParametricNonLinEq + SimpleSolver + HomotopySolver + CachingSolver                    */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    const ref_sub *sub;
    /* ParametricNonLinEq buffers (src/solvers.jl:6-36) */
    double *res, *Jp, *J, *pfull, *Jq, *q;
    /* SimpleSolver (src/solvers.jl:151-179) */
    double *z, *lu, *last_z, *last_p, *last_Jp, *last_lu, *tmp_nn, *tmp_np;
    int *ipiv, *last_ipiv;
    int iters;
    double resmaxabs, tol;
    int maxiter;
    /* HomotopySolver (src/solvers.jl:247-260) */
    double *start_p, *pa;
    int h_iters;
    /* CachingSolver (src/solvers.jl:319-339) */
    double *ps, *zs;
    int num_ps, cap_ps;
    int cache_limit, cache_head; /* > 0: bounded FIFO variant (the GPU's solution cache), 0: the reference's unbounded store */
} ref_solver;

/* set_p closure (src/ACME.jl:237-243): pfull <- q0 + pexp*p */
static void set_p(ref_solver *s, const double *p) {
    const ref_sub *b = s->sub;
    for (int i = 0; i < b->nq; ++i) s->pfull[i] = b->q0[i];
    for (int j = 0; j < b->np; ++j)
        for (int i = 0; i < b->nq; ++i) s->pfull[i] += b->pexp[i + (size_t)j * b->nq] * p[j];
}

/* residual closure (src/ACME.jl:178-188) with CircuitNLFunc (src/circuit.jl:10-17) */
static void evaluate(ref_solver *s, const double *z) {
    const ref_sub *b = s->sub;
    int nn = b->nn, nq = b->nq;
    for (int i = 0; i < nq; ++i) s->q[i] = s->pfull[i];
    for (int j = 0; j < nn; ++j)
        for (int i = 0; i < nq; ++i) s->q[i] += b->fq[i + (size_t)j * nq] * z[j];
    memset(s->Jq, 0, sizeof(double) * (size_t)nn * nq);
    for (int e = 0; e < b->n_elems; ++e) {
        const ref_elem *el = &b->elems[e];
        double r[4], J[32];
        acme_ref_eval_element(el->kind, el->par, s->q + el->qoff, r, J);
        for (int i = 0; i < el->nn; ++i) {
            s->res[el->roff + i] = r[i];
            for (int j = 0; j < el->nq; ++j) /* Jq column-major nn x nq */
                s->Jq[(el->roff + i) + (size_t)(el->qoff + j) * nn] = J[i * el->nq + j];
        }
    }
    /* J <- Jq*fq (dense gemm in the reference) */
    for (int j = 0; j < nn; ++j)
        for (int i = 0; i < nn; ++i) {
            double acc = 0.0;
            for (int c = 0; c < nq; ++c)
                acc += s->Jq[i + (size_t)c * nn] * b->fq[c + (size_t)j * nq];
            s->J[i + (size_t)j * nn] = acc;
        }
}

/* calc_Jp closure (src/ACME.jl:246-251): Jp <- Jq*pexp */
static void calc_Jp(ref_solver *s) {
    const ref_sub *b = s->sub;
    int nn = b->nn, nq = b->nq, np = b->np;
    for (int j = 0; j < np; ++j)
        for (int i = 0; i < nn; ++i) {
            double acc = 0.0;
            for (int c = 0; c < nq; ++c)
                acc += s->Jq[i + (size_t)c * nn] * b->pexp[c + (size_t)j * nq];
            s->Jp[i + (size_t)j * nn] = acc;
        }
}

static int simple_hasconverged(const ref_solver *s) { return s->resmaxabs < s->tol; }

/* set_extrapolation_origin(solver, p, z, Jp, linsolver) (src/solvers.jl:191-196) */
static void store_origin(ref_solver *s, const double *p, const double *z) {
    int nn = s->sub->nn, np = s->sub->np;
    memcpy(s->last_lu, s->lu, sizeof(double) * (size_t)nn * nn);
    memcpy(s->last_ipiv, s->ipiv, sizeof(int) * (size_t)nn);
    memcpy(s->last_Jp, s->Jp, sizeof(double) * (size_t)nn * np);
    memmove(s->last_p, p, sizeof(double) * (size_t)np);
    memmove(s->last_z, z, sizeof(double) * (size_t)nn);
}

/* set_extrapolation_origin(solver, p, z) (src/solvers.jl:183-189) */
static void set_origin(ref_solver *s, const double *p, const double *z) {
    int nn = s->sub->nn;
    set_p(s, p);
    evaluate(s, z);
    memcpy(s->lu, s->J, sizeof(double) * (size_t)nn * nn);
    acme_ref_lu_factor(nn, s->lu, s->ipiv);
    calc_Jp(s);
    store_origin(s, p, z);
}

/* solve(::SimpleSolver, p) (src/solvers.jl:207-236) */
static const double *simple_solve(ref_solver *s, const double *p) {
    const ref_sub *b = s->sub;
    int nn = b->nn, np = b->np;
    set_p(s, p);
    for (int i = 0; i < np; ++i) s->tmp_np[i] = p[i];
    for (int i = 0; i < np; ++i) s->tmp_np[i] += -1.0 * s->last_p[i];
    for (int i = 0; i < nn; ++i) s->tmp_nn[i] = 0.0;
    for (int j = 0; j < np; ++j)
        for (int i = 0; i < nn; ++i) s->tmp_nn[i] += s->last_Jp[i + (size_t)j * nn] * s->tmp_np[j];
    acme_ref_lu_solve(nn, s->last_lu, s->last_ipiv, s->tmp_nn);
    for (int i = 0; i < nn; ++i) s->z[i] = s->last_z[i];
    for (int i = 0; i < nn; ++i) s->z[i] += -1.0 * s->tmp_nn[i];

    for (s->iters = 1; s->iters <= s->maxiter; ++s->iters) {
        evaluate(s, s->z);
        double m = 0.0;
        int nanres = 0;
        for (int i = 0; i < nn; ++i) { /* maximum(abs, res): NaN propagates */
            double v = fabs(s->res[i]);
            if (isnan(v)) nanres = 1;
            if (v > m) m = v;
        }
        s->resmaxabs = nanres ? NAN : m;
        if (getenv("ACME_REF_TRACE") && atoi(getenv("ACME_REF_TRACE")) > 1) fprintf(stderr, "ref newton it %d resmax %g z0 %.17g\n", s->iters, s->resmaxabs, s->z[0]);
        int finite = isfinite(s->resmaxabs);
        if (finite)
            for (int i = 0; i < nn * nn; ++i)
                if (!isfinite(s->J[i])) { finite = 0; break; }
        if (!finite) return s->z;
        memcpy(s->lu, s->J, sizeof(double) * (size_t)nn * nn);
        if (!acme_ref_lu_factor(nn, s->lu, s->ipiv)) return s->z; /* J was singular */
        if (simple_hasconverged(s)) break;
        for (int i = 0; i < nn; ++i) s->tmp_nn[i] = s->res[i];
        acme_ref_lu_solve(nn, s->lu, s->ipiv, s->tmp_nn);
        for (int i = 0; i < nn; ++i) s->z[i] += -1.0 * s->tmp_nn[i];
    }
    if (s->iters > s->maxiter) s->iters = s->maxiter; /* Julia's loop variable ends at maxiter */
    if (simple_hasconverged(s)) {
        calc_Jp(s);
        store_origin(s, p, s->z);
    }
    return s->z;
}

/* solve(::CachingSolver, p) (src/solvers.jl:347-396), exhaustive-scan variant */
static const double *caching_solve(ref_solver *s, const double *p) {
    const ref_sub *b = s->sub;
    int nn = b->nn, np = b->np;
    double best = 0.0;
    for (int i = 0; i < np; ++i) best += (p[i] - s->last_p[i]) * (p[i] - s->last_p[i]);
    int idx = -1;
    for (int k = 0; k < s->num_ps; ++k) {
        double d = 0.0;
        for (int j = 0; j < np; ++j) {
            double t = s->ps[j + (size_t)k * np] - p[j];
            d += t * t;
        }
        if (d < best) {
            best = d;
            idx = k;
        }
    }
    if (idx >= 0) set_origin(s, s->ps + (size_t)idx * np, s->zs + (size_t)idx * nn);
    const double *z = simple_solve(s, p);
    if (s->iters > 5 && simple_hasconverged(s) && s->cache_limit > 0 && s->num_ps == s->cache_limit) {
        /* bounded variant: overwrite the oldest entry */
        memcpy(s->ps + (size_t)s->cache_head * np, p, sizeof(double) * (size_t)np);
        memcpy(s->zs + (size_t)s->cache_head * nn, z, sizeof(double) * (size_t)nn);
        s->cache_head = (s->cache_head + 1) % s->cache_limit;
    } else if (s->iters > 5 && simple_hasconverged(s)) {
        if (s->num_ps == s->cap_ps) {
            s->cap_ps = 2 * (s->num_ps + 1);
            s->ps = (double *)realloc(s->ps, sizeof(double) * (size_t)np * s->cap_ps + 8);
            s->zs = (double *)realloc(s->zs, sizeof(double) * (size_t)nn * s->cap_ps + 8);
        }
        memcpy(s->ps + (size_t)s->num_ps * np, p, sizeof(double) * (size_t)np);
        memcpy(s->zs + (size_t)s->num_ps * nn, z, sizeof(double) * (size_t)nn);
        ++s->num_ps;
    }
    return z;
}

static const double *base_solve(ref_solver *s, int kind, const double *p) {
    return kind == ACME_REF_SOLVER_HOMOTOPY_CACHING ? caching_solve(s, p) : simple_solve(s, p);
}

/* solve(::HomotopySolver, p) (src/solvers.jl:268-296) */
static const double *homotopy_solve(ref_solver *s, int kind, const double *p) {
    int np = s->sub->np;
    const double *z = base_solve(s, kind, p);
    s->h_iters = s->iters;
    if (getenv("ACME_REF_TRACE")) fprintf(stderr, "ref direct conv=%d its=%d resmax=%g\n", simple_hasconverged(s), s->iters, s->resmaxabs);
    if (!simple_hasconverged(s)) {
        double a = 0.5, best_a = 0.0;
        memcpy(s->start_p, s->last_p, sizeof(double) * (size_t)np);
        while (best_a < 1) {
            for (int i = 0; i < np; ++i) s->pa[i] = s->start_p[i];
            for (int i = 0; i < np; ++i) s->pa[i] *= (1 - a);
            for (int i = 0; i < np; ++i) s->pa[i] += a * p[i];
            z = base_solve(s, kind, s->pa);
            s->h_iters += s->iters;
            if (getenv("ACME_REF_TRACE")) fprintf(stderr, "ref homotopy a=%.17g best=%.17g conv=%d its=%d resmax=%g\n", a, best_a, simple_hasconverged(s), s->iters, s->resmaxabs);
            if (simple_hasconverged(s)) {
                best_a = a;
                a = 1.0;
            } else {
                double new_a = (a + best_a) / 2;
                if (!(best_a < new_a && new_a < a)) break;
                a = new_a;
            }
        }
    }
    return z;
}

/* ------------------------------------------------------------------------------------ */
/* runner                                                                                */
/* ------------------------------------------------------------------------------------ */
struct acme_ref_runner {
    const acme_ref_model *m;
    int solver_kind;
    ref_solver *solvers;
    double *x, *ucur, *ycur, *xnew, *z, *p; /* ModelRunner scratch (src/ACME.jl:570-585) */
};

static void solver_init(ref_solver *s, const ref_sub *b) {
    int nn = b->nn, nq = b->nq, np = b->np;
    memset(s, 0, sizeof *s);
    s->sub = b;
    s->res = dupd(NULL, nn); s->Jp = dupd(NULL, (size_t)nn * np); s->J = dupd(NULL, (size_t)nn * nn);
    s->pfull = dupd(NULL, nq); s->Jq = dupd(NULL, (size_t)nn * nq); s->q = dupd(NULL, nq);
    s->z = dupd(NULL, nn); s->lu = dupd(NULL, (size_t)nn * nn); s->last_z = dupd(NULL, nn);
    s->last_p = dupd(NULL, np); s->last_Jp = dupd(NULL, (size_t)nn * np);
    s->last_lu = dupd(NULL, (size_t)nn * nn); s->tmp_nn = dupd(NULL, nn); s->tmp_np = dupd(NULL, np);
    s->ipiv = (int *)calloc(nn ? nn : 1, sizeof(int));
    s->last_ipiv = (int *)calloc(nn ? nn : 1, sizeof(int));
    s->start_p = dupd(NULL, np); s->pa = dupd(NULL, np);
    s->tol = 1e-10; /* src/solvers.jl:175 */
    s->maxiter = 500; /* src/solvers.jl:207 */
    /* Solver(nleq, zeros(np), init_z) (src/ACME.jl:253-259; src/solvers.jl:164-178) */
    double *p0 = dupd(NULL, np);
    set_origin(s, p0, b->init_z);
    /* CachingSolver ctor (src/solvers.jl:327-333): the initial point is stored */
    s->cap_ps = 4;
    s->ps = dupd(NULL, (size_t)np * s->cap_ps + 1);
    s->zs = dupd(NULL, (size_t)nn * s->cap_ps + 1);
    memcpy(s->ps, p0, sizeof(double) * (size_t)np);
    memcpy(s->zs, b->init_z, sizeof(double) * (size_t)nn);
    s->num_ps = 1;
    free(p0);
}

static void solver_free(ref_solver *s) {
    free(s->res); free(s->Jp); free(s->J); free(s->pfull); free(s->Jq); free(s->q); free(s->z);
    free(s->lu); free(s->last_z); free(s->last_p); free(s->last_Jp); free(s->last_lu);
    free(s->tmp_nn); free(s->tmp_np); free(s->ipiv); free(s->last_ipiv); free(s->start_p);
    free(s->pa); free(s->ps); free(s->zs);
}

acme_ref_runner *acme_ref_runner_create(const acme_ref_model *m, int solver_kind) {
    if (solver_kind < 0 || solver_kind > 2) return NULL;
    acme_ref_runner *r = (acme_ref_runner *)calloc(1, sizeof *r);
    r->m = m;
    r->solver_kind = solver_kind;
    r->solvers = (ref_solver *)calloc(m->nsub ? m->nsub : 1, sizeof(ref_solver));
    int maxnp = 0;
    for (int i = 0; i < m->nsub; ++i) {
        solver_init(&r->solvers[i], &m->subs[i]);
        if (m->subs[i].np > maxnp) maxnp = m->subs[i].np;
    }
    r->x = dupd(NULL, m->nx); /* zeros(length(x0)), src/ACME.jl:145 */
    r->ucur = dupd(NULL, m->nu);
    r->ycur = dupd(NULL, m->ny);
    r->xnew = dupd(NULL, m->nx);
    r->z = dupd(NULL, m->nn_total);
    r->p = dupd(NULL, maxnp);
    return r;
}

void acme_ref_runner_destroy(acme_ref_runner *r) {
    if (!r) return;
    for (int i = 0; i < r->m->nsub; ++i) solver_free(&r->solvers[i]);
    free(r->solvers);
    free(r->x); free(r->ucur); free(r->ycur); free(r->xnew); free(r->z); free(r->p);
    free(r);
}

void acme_ref_set_resabstol(acme_ref_runner *r, double tol) {
    for (int i = 0; i < r->m->nsub; ++i) r->solvers[i].tol = tol;
}

void acme_ref_set_maxiter(acme_ref_runner *r, int maxiter) {
    for (int i = 0; i < r->m->nsub; ++i) r->solvers[i].maxiter = maxiter;
}

/* CachingSolver with a bounded FIFO store of `limit` solutions (0 = unbounded, the reference):
 * the variant the GPU kernel implements, for iteration-count parity tests.  Call before running. */
void acme_ref_set_cache_limit(acme_ref_runner *r, int limit) {
    for (int i = 0; i < r->m->nsub; ++i) {
        r->solvers[i].cache_limit = limit;
        r->solvers[i].cache_head = 0;
    }
}

static const double *any_solve(acme_ref_runner *r, int idx, const double *p, int *iters) {
    ref_solver *s = &r->solvers[idx];
    const double *z;
    if (r->solver_kind == ACME_REF_SOLVER_SIMPLE) {
        z = simple_solve(s, p);
        *iters = s->iters;
    } else {
        z = homotopy_solve(s, r->solver_kind, p);
        *iters = s->h_iters;
    }
    return z;
}

void acme_ref_solve(acme_ref_runner *r, int sub, const double *p, double *z, int *converged,
                    int *iters) {
    const double *zz = any_solve(r, sub, p, iters);
    memcpy(z, zz, sizeof(double) * (size_t)r->m->subs[sub].nn);
    *converged = simple_hasconverged(&r->solvers[sub]);
}

/* step! (src/ACME.jl:666-715); returns 1 on "got non-finite result" */
static int step(acme_ref_runner *r, const double *u, double *y, long long n,
                acme_ref_report *rep) {
    const acme_ref_model *m = r->m;
    int nx = m->nx, nu = m->nu, ny = m->ny, nnt = m->nn_total;
    for (int i = 0; i < nu; ++i) r->ucur[i] = u[(size_t)n * nu + i];
    int zoff = 0;
    for (int i = 0; i < nnt; ++i) r->z[i] = 0.0;
    for (int idx = 0; idx < m->nsub; ++idx) {
        const ref_sub *b = &m->subs[idx];
        double *p = r->p;
        for (int i = 0; i < b->np; ++i) p[i] = 0.0;
        for (int j = 0; j < nx; ++j) /* gemv 'N' dq*x */
            for (int i = 0; i < b->np; ++i) p[i] += b->dq[i + (size_t)j * b->np] * r->x[j];
        for (int j = 0; j < nu; ++j)
            for (int i = 0; i < b->np; ++i) p[i] += b->eq[i + (size_t)j * b->np] * r->ucur[j];
        if (idx > 0)
            for (int j = 0; j < nnt; ++j)
                for (int i = 0; i < b->np; ++i) p[i] += b->fqprev[i + (size_t)j * b->np] * r->z[j];
        int iters = 0;
        const double *zsub = any_solve(r, idx, p, &iters);
        if (rep) {
            rep->iters_total += iters;
            if (iters > rep->iters_max) rep->iters_max = iters;
        }
        if (!simple_hasconverged(&r->solvers[idx])) {
            int allfinite = 1;
            for (int i = 0; i < b->nn; ++i)
                if (!isfinite(zsub[i])) allfinite = 0;
            if (allfinite) { /* @warn and continue (:689-690) */
                if (rep) {
                    ++rep->n_warn;
                    if (rep->first_nonconverged < 0) rep->first_nonconverged = n;
                }
            } else { /* error(...) (:691-692) */
                if (rep && rep->first_nonfinite < 0) rep->first_nonfinite = n;
                return 1;
            }
        }
        for (int i = 0; i < b->nn; ++i) r->z[zoff + i] = zsub[i];
        zoff += b->nn;
    }
    if (ny > 0) { /* y from the OLD x (:699-706) */
        for (int i = 0; i < ny; ++i) r->ycur[i] = m->y0[i];
        for (int j = 0; j < nx; ++j)
            for (int i = 0; i < ny; ++i) r->ycur[i] += m->dy[i + (size_t)j * ny] * r->x[j];
        for (int j = 0; j < nu; ++j)
            for (int i = 0; i < ny; ++i) r->ycur[i] += m->ey[i + (size_t)j * ny] * r->ucur[j];
        for (int j = 0; j < nnt; ++j)
            for (int i = 0; i < ny; ++i) r->ycur[i] += m->fy[i + (size_t)j * ny] * r->z[j];
        for (int i = 0; i < ny; ++i) y[(size_t)n * ny + i] = r->ycur[i];
    }
    if (nx > 0) { /* x update (:708-714) */
        for (int i = 0; i < nx; ++i) r->xnew[i] = m->x0[i];
        for (int j = 0; j < nx; ++j)
            for (int i = 0; i < nx; ++i) r->xnew[i] += m->a[i + (size_t)j * nx] * r->x[j];
        for (int j = 0; j < nu; ++j)
            for (int i = 0; i < nx; ++i) r->xnew[i] += m->b[i + (size_t)j * nx] * r->ucur[j];
        for (int j = 0; j < nnt; ++j)
            for (int i = 0; i < nx; ++i) r->xnew[i] += m->c[i + (size_t)j * nx] * r->z[j];
        memcpy(r->x, r->xnew, sizeof(double) * (size_t)nx);
    }
    return 0;
}

int acme_ref_run(acme_ref_runner *r, const double *u, double *y, long long T,
                 acme_ref_report *rep) {
    if (rep) {
        memset(rep, 0, sizeof *rep);
        rep->first_nonconverged = -1;
        rep->first_nonfinite = -1;
    }
    long long s0 = g_lu_swaps, c0 = g_lu_count;
    int rc = 0;
    for (long long n = 0; n < T; ++n) {
        if (step(r, u, y, n, rep)) {
            rc = 1;
            break;
        }
    }
    if (rep) {
        rep->lu_swaps = g_lu_swaps - s0;
        rep->lu_count = g_lu_count - c0;
    }
    return rc;
}

void acme_ref_get_x(const acme_ref_runner *r, double *x) {
    memcpy(x, r->x, sizeof(double) * (size_t)r->m->nx);
}
void acme_ref_set_x(acme_ref_runner *r, const double *x) {
    memcpy(r->x, x, sizeof(double) * (size_t)r->m->nx);
}
void acme_ref_get_origin(const acme_ref_runner *r, int sub, double *p, double *z) {
    const ref_solver *s = &r->solvers[sub];
    memcpy(p, s->last_p, sizeof(double) * (size_t)s->sub->np);
    memcpy(z, s->last_z, sizeof(double) * (size_t)s->sub->nn);
}
void acme_ref_set_origin(acme_ref_runner *r, int sub, const double *p, const double *z) {
    set_origin(&r->solvers[sub], p, z);
}
