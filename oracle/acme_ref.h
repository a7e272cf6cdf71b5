/* acme_ref.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's per-sample hot path, used only as the
 * checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The
 * product path (acme_jl_amd + libacme_hip.so) must never call into this library.
 *
 * PARITY PIN STATUS: the reference is Julia-only and cannot run in this environment, so
 * this restatement is pinned against (a) the two doctest golden vectors of the
 * reference (docs/src/gettingstarted.md:106-113, docs/src/ug.md:107-114), (b) the
 * analytic known-answer tests of test/runtests.jl (:23-41, :70-86, :170-183, :207-219,
 * :267-292, :386-429, :489-546, :590-662) replayed in tests/, and (c) the model-size
 * pins np(model,k) (:699,:724,:734,:744,:757-759,:768,:777,:788-791).  Sample values of
 * superover/birdie outputs are NOT pinned by the reference itself (it only asserts
 * size(y)); for those, parity is "unpinned" beyond (a)-(c).
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * ACME.jl tree).
 */
#ifndef ACME_REF_H
#define ACME_REF_H

#ifdef __cplusplus
extern "C" {
#endif

/* element kinds (same numbering as acme_jl_amd/circuit.py and include/acme_hip.h) */
enum {
    ACME_REF_KIND_DIODE = 1,  /* par: is, eta                      src/elements.jl:236-245 */
    ACME_REF_KIND_BJT = 2,    /* par: ise,isc,etae,etac,bf,br,ile,ilc,etael,etacl,vaf,var,ikf,ikr
                                                                   src/elements.jl:309-406 */
    ACME_REF_KIND_POT = 3,    /* par: r                            src/elements.jl:20-31   */
    ACME_REF_KIND_MOSFET = 4, /* par: pol,lambda,nvt,vt[4],nalpha,alpha[4]
                                                                   src/elements.jl:436-481 */
    ACME_REF_KIND_MACAK = 5,  /* par: gain, scale                  src/elements.jl:536-551 */
    ACME_REF_KIND_JA = 6      /* par: Ms,a,alpha,c,k               src/elements.jl:100-135 */
};
#define ACME_REF_MAX_PAR 16

/* solver stacks (third positional argument of DiscreteModel, src/ACME.jl:150) */
enum {
    ACME_REF_SOLVER_SIMPLE = 0,          /* SimpleSolver                       */
    ACME_REF_SOLVER_HOMOTOPY = 1,        /* HomotopySolver{SimpleSolver}       */
    ACME_REF_SOLVER_HOMOTOPY_CACHING = 2 /* HomotopySolver{CachingSolver{SimpleSolver}} (default) */
};

typedef struct acme_ref_model acme_ref_model;
typedef struct acme_ref_runner acme_ref_runner;

/* run report, also the vehicle for the reference's failure semantics
 * (src/ACME.jl:688-694): n_warn counts "Failed to converge" warnings; first_nonfinite
 * is the 0-based sample index at which the reference would have thrown, or -1. */
typedef struct {
    long long n_warn;
    long long first_nonconverged; /* 0-based sample, -1 if none */
    long long first_nonfinite;    /* 0-based sample, -1 if none */
    long long iters_total;        /* sum of needediterations over samples and sub-problems */
    long long iters_max;          /* max needediterations of any single solve */
    long long lu_swaps;           /* diagnostic: row interchanges performed in setlhs! */
    long long lu_count;           /* diagnostic: number of factorisations */
} acme_ref_report;

/* --- model (struct DiscreteModel, src/ACME.jl:118-148); all matrices column-major --- */
acme_ref_model *acme_ref_model_create(int nx, int nu, int ny, int nn_total,
                                      const double *a, const double *b, const double *c,
                                      const double *x0, const double *dy, const double *ey,
                                      const double *fy, const double *y0);
/* append one nonlinear sub-problem; elem_par is n_elems x ACME_REF_MAX_PAR row-major */
int acme_ref_model_add_subproblem(acme_ref_model *m, int nn, int nq, int np,
                                  const double *pexp, const double *dq, const double *eq,
                                  const double *fqprev, const double *fq, const double *q0,
                                  const double *init_z, int n_elems, const int *elem_kind,
                                  const int *elem_qoff, const int *elem_roff,
                                  const double *elem_par);
void acme_ref_model_destroy(acme_ref_model *m);

/* --- runner = ModelRunner + the model's mutable state (x, solver state) --------------- */
acme_ref_runner *acme_ref_runner_create(const acme_ref_model *m, int solver_kind);
void acme_ref_runner_destroy(acme_ref_runner *r);
void acme_ref_set_resabstol(acme_ref_runner *r, double tol);   /* src/solvers.jl:181 */
void acme_ref_set_maxiter(acme_ref_runner *r, int maxiter);    /* src/solvers.jl:207 */
/* CachingSolver with a bounded FIFO store (0 = unbounded = the reference); the GPU's variant */
void acme_ref_set_cache_limit(acme_ref_runner *r, int limit);

/* run!(runner, y, u) (src/ACME.jl:650-664): u is nu x T, y is ny x T, column-major.
 * Returns 0, or 1 if the reference would have thrown "got non-finite result" (then
 * report->first_nonfinite is the sample; y beyond it is untouched, x is left at it). */
int acme_ref_run(acme_ref_runner *r, const double *u, double *y, long long T,
                 acme_ref_report *report);

/* state access: x (nx) and, per sub-problem, the extrapolation origin (last_p, last_z)
 * (get/set_extrapolation_origin, src/solvers.jl:183-198) */
void acme_ref_get_x(const acme_ref_runner *r, double *x);
void acme_ref_set_x(acme_ref_runner *r, const double *x);
void acme_ref_get_origin(const acme_ref_runner *r, int sub, double *p, double *z);
void acme_ref_set_origin(acme_ref_runner *r, int sub, const double *p, const double *z);

/* plugin-level entry: solve(solver, p) + hasconverged + needediterations
 * (src/solvers.jl:207-236, 268-302) on sub-problem `sub` */
void acme_ref_solve(acme_ref_runner *r, int sub, const double *p, double *z, int *converged,
                    int *iters);

/* LinearSolver (src/solvers.jl:38-132) for unit tests: factors n x n column-major in
 * place, ipiv 0-based; returns 1 on success, 0 if singular */
int acme_ref_lu_factor(int n, double *factors, int *ipiv);
void acme_ref_lu_solve(int n, const double *factors, const int *ipiv, double *x);

/* element function (src/elements.jl closures): res[nn], Jq[nn*nq] row-major */
void acme_ref_eval_element(int kind, const double *par, const double *q, double *res,
                           double *Jq);

#ifdef __cplusplus
}
#endif
#endif
