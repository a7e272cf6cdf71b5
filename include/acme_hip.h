/* acme_hip.h -- C ABI of libacme_hip.so: batched, MI355X-native run!(::DiscreteModel, u).
 *
 * This is the drop-in boundary for the one hot path of ACME.jl that this project
 * accelerates.  Everything is plain C: pointers, sizes, integer status codes; no C++
 * or torch types.  A maintainer binds it from Julia with `ccall` (see INTEGRATION.md for
 * the glue that implements ModelRunner/run! on top of it); this repository's own host
 * side binds it from Python with ctypes (acme_jl_amd/runner.py).
 *
 * Reference interfaces replaced (file:line relative to the ACME.jl tree):
 *   DiscreteModel data .............................. src/ACME.jl:118-148
 *   closures func/set_p/calc_Jp -> element table .... src/ACME.jl:176-194,236-252,
 *                                                     src/circuit.jl:6-20,68-86
 *   ModelRunner(model, showprogress) ................ src/ACME.jl:570-604
 *   run!(runner, y, u) / step! ...................... src/ACME.jl:650-715
 *   solver plugin contract (set_resabstol!, get/set_extrapolation_origin,
 *     hasconverged, needediterations) ............... src/solvers.jl:181-205,262-302
 *
 * Conventions
 *   * All matrices are column-major Float64, exactly Julia's Matrix{Float64} layout.
 *   * A batch holds N independent instances of one model.  u is [N][T][nu] and y is
 *     [N][T][ny] doubles, i.e. instance i's block is the reference's nu x T (ny x T)
 *     column-major matrix, blocks back to back; a batch of one is bit-layout identical
 *     to the reference's u and y.
 *   * Every entry point returns ACME_OK (0) or a negative error code and never unwinds;
 *     acme_last_error() returns a thread-local message for the last failure.
 *   * Threading: handles are single-owner; distinct handles are independent.
 *   * Devices: a batch lives on one HIP device (acme_options.device); every entry point switches
 *     to it for the duration of the call and restores the caller's current device on return.
 *   * Failure semantics of step! (src/ACME.jl:688-694) are reported per instance in
 *     acme_report: n_warn counts "Failed to converge" warnings; first_nonfinite >= 0 is
 *     the sample at which the reference would have thrown -- that instance stops
 *     advancing there (its x stays at that sample, y from there on is NaN).
 */
#ifndef ACME_HIP_H
#define ACME_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ACME_OK 0
#define ACME_ERR_INVALID (-1)     /* bad argument / DimensionMismatch */
#define ACME_ERR_UNSUPPORTED (-2) /* model does not fit the kernel (see message) */
#define ACME_ERR_HIP (-3)         /* HIP runtime failure */
#define ACME_ERR_NO_DEVICE (-4)   /* no usable GPU: the product path has no CPU fallback */

/* element kinds of the nonlinear element table and their parameter vectors
 * (ACME_MAX_ELEM_PAR doubles per element, unused entries zero) */
#define ACME_KIND_DIODE 1  /* is, eta                                   src/elements.jl:236-245 */
#define ACME_KIND_BJT 2    /* ise,isc,etae,etac,bf,br,ile,ilc,etael,etacl,vaf,var,ikf,ikr
                                                                       src/elements.jl:309-406 */
#define ACME_KIND_POT 3    /* r                                         src/elements.jl:20-31   */
#define ACME_KIND_MOSFET 4 /* polarity,lambda,nvt,vt[4],nalpha,alpha[4] src/elements.jl:436-481
                              (the reference takes polynomials of any length, :436-450; here 1 ... 4 coefficients
                              each: acme_model_add_subproblem returns ACME_ERR_UNSUPPORTED beyond that) */
#define ACME_KIND_MACAK 5  /* gain, scale                               src/elements.jl:536-551 */
#define ACME_KIND_JA 6     /* Ms,a,alpha,c,k                            src/elements.jl:100-135 */
#define ACME_MAX_ELEM_PAR 16

/* solver selection = third positional argument of DiscreteModel (src/ACME.jl:150) */
#define ACME_SOLVER_SIMPLE 0   /* SimpleSolver                  src/solvers.jl:151-236 */
#define ACME_SOLVER_HOMOTOPY 1 /* HomotopySolver{SimpleSolver}  src/solvers.jl:247-302 */
/* HomotopySolver{CachingSolver{SimpleSolver}} (src/solvers.jl:303-405, the reference's default
 * stack) with a BOUNDED store: per instance and sub-problem the last 16 stored solutions, first in
 * first out, instead of the reference's ever-growing k-d tree.  Same lookup rule (a stored p
 * strictly nearer than the current extrapolation origin becomes the origin) and storing rule
 * (converged base solve that needed more than 5 iterations). */
#define ACME_SOLVER_CACHING_HOMOTOPY 2

/* where u / y live */
#define ACME_MEM_HOST 0
#define ACME_MEM_DEVICE 1

typedef struct acme_model acme_model;
typedef struct acme_batch acme_batch;

typedef struct {
    int solver;    /* ACME_SOLVER_*; default HOMOTOPY */
    double tol;    /* residual max-abs tolerance, default 1e-10 (src/solvers.jl:175) */
    int maxiter;   /* Newton iterations per base solve, default 500 (src/solvers.jl:207) */
    int device;    /* HIP device ordinal, -1 = current device */
    int per_instance_matrices; /* 0: all instances share the model's matrices;
                                  1: every instance has its own (acme_batch_set_matrices) */
} acme_options;

typedef struct {
    long long n_warn;             /* "Failed to converge" warnings so far            */
    long long first_nonconverged; /* 0-based sample of the first warning, -1 if none */
    long long first_nonfinite;    /* 0-based sample of the fatal error, -1 if none   */
    long long iters_total;        /* sum of needediterations over all samples        */
    long long iters_max;          /* max needediterations of one sample              */
} acme_report;

const char *acme_last_error(void);
/* number of visible HIP devices (0 if none / runtime unavailable) */
int acme_device_count(void);
void acme_default_options(acme_options *opts);

/* ---- model: the data of struct DiscreteModel (src/ACME.jl:118-148) ------------------ */
int acme_model_create(int nx, int nu, int ny, int nn_total, const double *a, const double *b,
                      const double *c, const double *x0, const double *dy, const double *ey,
                      const double *fy, const double *y0, acme_model **out);
/* one nonlinear sub-problem: pexps/dqs/eqs/fqprevs/fqs/q0s[idx] (src/ACME.jl:123-128), the
 * initial extrapolation origin z (p = 0; src/ACME.jl:253-259) and the element table in
 * CircuitNLFunc order (src/circuit.jl:68-86); elem_par is n_elems x ACME_MAX_ELEM_PAR
 * row-major */
int acme_model_add_subproblem(acme_model *m, int nn, int nq, int np, const double *pexp,
                              const double *dq, const double *eq, const double *fqprev,
                              const double *fq, const double *q0, const double *init_z,
                              int n_elems, const int *elem_kind, const int *elem_qoff,
                              const int *elem_roff, const double *elem_par);
/* optional performance hint: order[pos] = residual row (0-based, in element-table order)
 * handled by lane `pos` when a batch is created.  The elimination keeps the row sitting in pivot
 * position while every multiplier satisfies |l| <= 8 (threshold partial pivoting; the
 * reference's setlhs!, src/solvers.jl:58-78, is the threshold 1) and only otherwise re-learns
 * the order with the reference's first-strict-maximum search; listing the equations in their
 * usual pivot order spares the first such searches.  Either way the factorisation is a valid LU
 * of the same Jacobian: results of different row orders agree to rounding.  n = 0 restores the
 * natural order. */
int acme_model_set_row_order(acme_model *m, int sub, const int *order, int n);
void acme_model_destroy(acme_model *m);
/* dims = {nn, nq, np, nx, nu, ny} of the instantiated kernel shape the model runs in */
int acme_model_kernel_shape(const acme_model *m, int dims[6]);
/* which kernel variant the model runs in: *condensed_rows = number of residual rows eliminated ahead of
 * the Newton iteration (the potentiometer rows of DESIGN.md 2 "Round 4"; 0 = none), *generic = 0 for a tuned
 * shape; 2 when no instantiated shape holds the model and the cooperative run-time-sized kernel takes it (one
 * sub-problem of up to 64 unknowns, working arrays in LDS: csrc/acme_coop.h); 1 for the lane-per-instance kernel
 * that takes everything else (csrc/acme_generic.h).  Either pointer may be null. */
int acme_model_kernel_variant(const acme_model *m, int *condensed_rows, int *generic);

/* ---- batch: N instances + their device-resident mutable state ----------------------- */
/* state starts like a fresh DiscreteModel: x = 0 (src/ACME.jl:145), origin (0, init_z) */
int acme_batch_create(const acme_model *m, long long n_instances, const acme_options *opts,
                      acme_batch **out);
void acme_batch_destroy(acme_batch *b);
/* which kernel variant THIS BATCH runs in, as acme_model_kernel_variant reports it for a model (*family: 0 tuned shape, 1
 * lane-per-instance generic kernel, 2 cooperative mid-size kernel).  It can differ from its model's: instances with element
 * parameters of their own (acme_batch_set_matrices) move a batch off the condensed shape (the library rebuilds it on the
 * plain shape by itself: the reference's models each carry their own closures, src/elements.jl:236-245,309-406) and a
 * mid-size batch to the lane-per-instance kernel.  Either pointer may be null. */
int acme_batch_kernel_variant(const acme_batch *b, int *condensed_rows, int *family);

/* per-instance matrices (Monte-Carlo component tolerances, sweeps over element parameters): instance i uses the
 * matrices AND the element closures' parameters of models[i] -- in the reference every model carries its own element
 * closures (src/elements.jl:236-245, 309-406).  All models must have the batch model's dimensions and circuit
 * STRUCTURE (the same element kinds in the same residual rows reading the same q entries); the element parameters
 * -- a diode's is / eta, a transistor's betas, which Gummel-Poon refinements it has -- may differ: the batch then
 * keeps one element table per instance (a block of the 16-lane kernels stages its 16 tables in LDS; if they do not
 * fit a compute unit the call fails with ACME_ERR_UNSUPPORTED).  The instances take the freshly constructed state of their model (x = 0, each
 * solver's extrapolation origin at p = 0, z = the model's init_z; src/ACME.jl:145,253-259).
 * Only valid for batches created with per_instance_matrices = 1. */
int acme_batch_set_matrices(acme_batch *b, long long first, long long count,
                            const acme_model *const *models);

/* run!(runner, y, u): advance every instance by T samples (src/ACME.jl:650-664).
 * mem = ACME_MEM_HOST: u/y are host buffers, copied in time slices that overlap the kernel.  On a batch with
 *   acme_batch_set_host_retention (arrays page-locked and mapped), runs of 4096+ samples are STREAMED: one launch;
 *   the kernel writes y to the caller's array itself and reads u from an HBM staging buffer the copy engine fills
 *   while the kernel runs (waves that get ahead of the copy wait) -- the device-resident rate; not with a progress
 *   callback installed, for memory that cannot be page-locked, for batches of more blocks than the chip holds at
 *   once, nor for the lane-per-instance and generic kernels (time slices).  The call returns when y is complete;
 * mem = ACME_MEM_DEVICE: u/y are device pointers on the batch's device and `stream` is the
 * hipStream_t to launch on (NULL = default stream); the call is then asynchronous. */
int acme_batch_run(acme_batch *b, const double *u, double *y, long long T, int mem,
                   void *stream);
/* run! with CONSTANT input rows (src/ACME.jl:672-674 copies column n of u into ucur sample by sample: a row that never
 * changes -- a potentiometer position, a supply voltage, a mix control of a parameter sweep -- need not be materialised
 * T times).  const_mask: bit k set = input row k keeps the value u_const[i * nu + k] (u_const: [N][nu], the entries of
 * the other rows are ignored) for the whole call; u_var then holds only the rows whose bit is clear, in row order:
 * [N][T][nu_var] with nu_var = nu - popcount(const_mask).  The full input rows are put together ON THE DEVICE, time
 * slice by time slice (HBM traffic the kernels do not notice), so a host-buffer run moves nu_var / nu of the bytes
 * over the bus -- the headline sweep (three pot rows of four inputs): 2.9 instead of 11.6 GB per second of audio.
 * Results are those of acme_batch_run on the materialised u, bit for bit.  mem / stream as acme_batch_run (u_var and
 * u_const live where mem says); y: [N][T][ny].  Host arrays: time slices copied in and out beside the kernels; with
 * acme_batch_set_host_retention u_var and y are page-locked once and kept (headline: 0.92 x the device-resident rate). */
int acme_batch_run_const(acme_batch *b, const double *u_var, const double *u_const, unsigned long long const_mask,
                         double *y, long long T, int mem, void *stream);
/* Host-buffer runs and page-locking.  By DEFAULT the library never keeps anything of the caller's arrays beyond the
 * call: u and y are copied from / to ordinary (pageable) memory in time slices that overlap the kernel -- the right
 * thing for one-shot calls and for wrappers whose arrays are per-call temporaries or garbage-collected (locking
 * 14.4 GB costs more than running them: headline 0.39 s this way, 0.70 s locked and streamed).
 * keep != 0: the caller PROMISES that the arrays it passes stay allocated until it passes others, calls
 * acme_batch_release_host_buffers or destroys the batch.  The library then page-locks and maps them
 * (hipHostRegister) on first use and keeps the last range of each direction locked: runs of 4096+ samples are
 * STREAMED (see acme_batch_run) at the device-resident rate -- what the in-place run!(runner, y, u) with reused
 * arrays is for (src/ACME.jl:650-664).  Memory that cannot be locked is copied from as it is.
 * ACME_HOST_REGISTER=0 in the environment disables the locking altogether. */
int acme_batch_set_host_retention(acme_batch *b, int keep);
/* un-page-lock what a retaining batch (above) holds: before the caller frees or resizes its arrays while the
 * batch lives on.  Idempotent. */
int acme_batch_release_host_buffers(acme_batch *b);
/* @showprogress of run!(runner, y, u) (src/ACME.jl:587-604,653): `fn(user, samples_done, samples_total)`
 * is called on the calling thread (the worker thread of an asynchronous run) after every time slice of a
 * host-buffer run -- 8 to 24 per run of 4096+ samples; a callback makes such a run sliced rather than streamed,
 * at ~0.91 of the speed -- and once at the end of any other run.  fn = NULL removes it.  The callback must not
 * call into the batch. */
typedef void (*acme_progress_fn)(void *user, long long samples_done, long long samples_total);
int acme_batch_set_progress_callback(acme_batch *b, acme_progress_fn fn, void *user);

/* acme_batch_run without blocking the caller: the same run on a worker thread of the library (a
 * host-buffer run drives its time-slice pipeline from there).  ONE host thread can thereby keep one
 * batch per GPU of a node busy -- start all, then acme_batch_wait each -- which is how a single
 * (Julia) process uses the 8 GPUs of a node without any collective: contiguous instance ranges, every
 * batch reading and writing its own slice of the caller's u / y.  At most one run is in flight per
 * batch; every other entry point taking the batch first waits for it.  u / y must stay valid until
 * acme_batch_wait, which returns the run's status (acme_last_error() then holds its message). */
int acme_batch_run_async(acme_batch *b, const double *u, double *y, long long T, int mem,
                         void *stream);
int acme_batch_wait(acme_batch *b);

/* Isolation of slow instances.  A launch lasts as long as its slowest wave, and every wave is one serial recurrence
 * over the samples: ONE pathological cell of a sweep -- say a potentiometer at an end stop that makes the model
 * singular, where the solver stack fails after ~900 Newton iterations per sample while every other cell needs 3 --
 * holds the results of all instances back by its own, hundreds of times longer, run.  With a threshold > 0 the
 * instances that needed more than `iters_per_sample` Newton iterations per sample over the batch's previous run are
 * launched on their own, on a stream of the library's; the others run on the caller's stream as if the slow ones were
 * not there.  What an instance computes does not depend on the group it runs in (bit-identical results).  A
 * device-pointer run (ACME_MEM_DEVICE) then completes on the caller's stream for the FAST instances only;
 * acme_batch_wait (or any entry point that reads the batch) completes the slow ones.  Host-buffer runs return
 * complete, as ever.  The first run of a batch is never split (nothing is known yet); groups are re-formed between
 * runs while no launch of the slow group is in flight.  0 switches the isolation off (default).  Not for batches the
 * lane-per-instance or generic kernels run. */
int acme_batch_set_isolation(acme_batch *b, double iters_per_sample);

/* Placement of the waves by their measured cost.  Two blocks of a launch share a compute unit and a launch ends with
 * its slowest SIMD; before a launch (at most once per 4 096 samples) two small kernels on the launch's own stream rank
 * the waves -- groups of 4 consecutive instances -- by the Newton iterations they needed since the last placement and
 * deal them to the launch's slots so that heavy and light waves share a SIMD (headline grid: 1.6 % over seconds 1-4 of a
 * signal, when the cells' costs differ most; nothing in the long steady state; tools/balance_probe.py).  No host synchronisation; what an instance computes does not depend on where it runs
 * (bit-identical results).  mode: -1 = the library decides (default: on when the launch has more waves than one round
 * of blocks holds), 0 = off, 1 = on.  Not for batches the lane-per-instance or generic kernels run, nor
 * while acme_batch_set_isolation is in force. */
int acme_batch_set_balance(acme_batch *b, int mode);
/* diagnostics: the placement the last launch used -- *n_slots instance slots (a multiple of 4: a wave has four), slot ->
 * instance in slot_to_instance[*n_slots], -1 for a slot left empty (an incomplete last wave's); at most 4 N slots; the
 * identity over N slots while nothing has been placed.
 * Either pointer may be null.  Completes the batch's outstanding work first. */
int acme_batch_get_placement(acme_batch *b, int *slot_to_instance, long long *n_slots);

/* The solver plugin contract, batched (src/solvers.jl:207-236, 268-302): for every instance
 *   z = solve(solver, p); converged = hasconverged(solver); iters = needediterations(solver)
 * on sub-problem `sub` (0-based): p is [N][np_sub], z is [N][nn_sub], converged/iters are [N].  Like the reference's
 * solver objects the call uses and updates the instance's extrapolation origin; x is not
 * touched and nothing is added to the run reports.  mem/stream as for acme_batch_run. */
int acme_batch_solve(acme_batch *b, int sub, const double *p, double *z, int *converged,
                     int *iters, int mem, void *stream);

/* get_extrapolation_jacobian(solver) (src/solvers.jl:198-201; what linearize builds the small-signal
 * model from, :407-414) for every instance: jac is [N][np_sub][nn_sub], i.e. per instance the
 * nn x np matrix  -(J \ Jp)  = dz/dp at the instance's extrapolation origin (last_p, last_z) of
 * sub-problem `sub`, column-major like Julia's Matrix{Float64}; NaN where J is singular there.  The
 * origin is re-linearised on the device (set_extrapolation_origin, :183-196, with the columns of Jp
 * riding along in the elimination); the batch's state is not modified.  mem/stream as for
 * acme_batch_run. */
int acme_batch_get_extrapolation_jacobian(acme_batch *b, int sub, double *jac, int mem, void *stream);

/* milliseconds the last acme_batch_run kernel took on the device (HIP events recorded on
 * the launch stream); synchronises with that launch */
int acme_batch_last_kernel_ms(acme_batch *b, float *ms);
/* accumulated device time and count of all launches since the last reset (HIP events on
 * the launch stream around every kernel); synchronises with the pending launches */
int acme_batch_kernel_time(acme_batch *b, double *ms_total, long long *launches, int reset);

/* per-instance reports, reports[n_instances]; synchronises */
int acme_batch_get_report(acme_batch *b, acme_report *reports);
int acme_batch_reset_report(acme_batch *b);

/* set_resabstol! (src/solvers.jl:181,262) */
int acme_batch_set_resabstol(acme_batch *b, double tol);

/* model.x and get/set_extrapolation_origin (src/solvers.jl:183-198) for all instances:
 * x is [N][nx], p is [N][sum np_k], z is [N][sum nn_k] (sub-problems concatenated in order);
 * NULL pointers are skipped */
int acme_batch_get_state(acme_batch *b, double *x, double *p, double *z);
int acme_batch_set_state(acme_batch *b, const double *x, const double *p, const double *z);

#ifdef __cplusplus
}
#endif
#endif
