# ACMEHip.jl -- Julia binding of libacme_hip.so (include/acme_hip.h) for ACME.jl.
#
# Drops the MI355X batch runner in behind ACME's own circuit-derivation front end WITHOUT any change
# to ACME: everything the C ABI needs is read from an ordinary `DiscreteModel` --
#   * the model matrices are its fields (src/ACME.jl:118-148),
#   * the element table (kind, parameters, q offset, residual offset per nonlinear element, in
#     `CircuitNLFunc` order, src/circuit.jl:68-86) is recovered by walking the closures the model
#     already holds: Julia closures expose their captured variables as fields, so
#     `model.nonlinear_eq_funcs[idx]` (src/ACME.jl:176-189) -> `circ_nl_func::CircuitNLFunc` ->
#     `.fs[k]` (src/circuit.jl:76-80: captures `q_indices`, `nleqfunc`) -> the element's own closure
#     (src/elements.jl), whose captured variables are the element parameters,
#   * each solver's initial extrapolation origin (p = 0, z = init_z; src/ACME.jl:253-259) comes from
#     `get_extrapolation_origin` of the innermost solver of `model.solvers[idx]`.
#
# Two levels, as SURVEY.md 8(b) describes:
#   * `BatchRunner` / `run!(::BatchRunner, y, u)` -- the performant seam: N instances of a model
#     advance together on the GPU (the batch analogue of `ModelRunner`, src/ACME.jl:570-664).
#   * `GPUBatchSolver <: ACME.NonlinearSolver` -- the solver plugin contract of src/solvers.jl:139-205
#     (constructor `S(nleq, initial_p, initial_z)`, `solve`, `hasconverged`, `needediterations`,
#     `set_resabstol!`, `get/set_extrapolation_origin`, `get_extrapolation_jacobian`) on top of
#     `acme_batch_solve`, so that `DiscreteModel(circ, t, ACMEHip.GPUBatchSolver)`, `steadystate`,
#     `linearize` and ACME's own solver tests run against the device code.  One `ccall` per solve:
#     this is for API parity and validation, not for throughput.
#
# NOTE: the build image of this repository has no Julia; this file is written against the ACME.jl
# sources cited above (and Julia >= 1.6 semantics) but has not been executed here.  The same ABI is
# exercised end to end by acme_jl_amd/runner.py (ctypes) and examples/abi_demo.c (plain C).
module ACMEHip

using ACME
using ACME: DiscreteModel, NonlinearSolver, ParametricNonLinEq
using LinearAlgebra: I
import ProgressMeter
import ACME: run!, solve, hasconverged, needediterations, set_resabstol!,
             get_extrapolation_origin, set_extrapolation_origin, get_extrapolation_jacobian

export BatchRunner, MultiBatchRunner, GPUBatchSolver, element_table, retain_host_buffers!, release_host_buffers!, set_isolation!, set_balance!

const lib = get(ENV, "ACME_HIP_LIB", "libacme_hip.so")

# ---- include/acme_hip.h ------------------------------------------------------------------------
const ACME_KIND_DIODE, ACME_KIND_BJT, ACME_KIND_POT = Cint(1), Cint(2), Cint(3)
const ACME_KIND_MOSFET, ACME_KIND_MACAK, ACME_KIND_JA = Cint(4), Cint(5), Cint(6)
const ACME_MAX_ELEM_PAR = 16
const ACME_SOLVER_SIMPLE, ACME_SOLVER_HOMOTOPY, ACME_SOLVER_CACHING_HOMOTOPY = Cint(0), Cint(1), Cint(2)
const ACME_MEM_HOST, ACME_MEM_DEVICE = Cint(0), Cint(1)
const KIND_NQ = Dict(1 => 2, 2 => 4, 3 => 5, 4 => 3, 5 => 2, 6 => 4)
const KIND_NN = Dict(1 => 1, 2 => 2, 3 => 2, 4 => 1, 5 => 1, 6 => 1)

struct AcmeOptions
    solver::Cint
    tol::Cdouble
    maxiter::Cint
    device::Cint
    per_instance_matrices::Cint
end

struct AcmeReport
    n_warn::Clonglong
    first_nonconverged::Clonglong
    first_nonfinite::Clonglong
    iters_total::Clonglong
    iters_max::Clonglong
end

lasterror() = unsafe_string(ccall((:acme_last_error, lib), Cstring, ()))
check(rc) = rc < 0 ? error("libacme_hip: " * lasterror()) : rc
"number of HIP devices the library sees"
device_count() = Int(ccall((:acme_device_count, lib), Cint, ()))

# ---- closures -> element table -----------------------------------------------------------------
"captured variable `name` of closure `f` (unwrapping the Box of a re-assigned capture)"
function captured(f, name::Symbol)
    v = getfield(f, name)
    return v isa Core.Box ? v.contents : v
end
hascaptured(f, name::Symbol) = name in fieldnames(typeof(f))
capturednames(f) = fieldnames(typeof(f))

"""
    describe_element(f) -> (kind, params)

`f` is the `nonlinear_eq` closure of one element (src/elements.jl); the kind is recognised by the
set of variables it captures, the parameter vector is laid out as include/acme_hip.h documents.
"""
function describe_element(f)
    names = capturednames(f)
    par = zeros(Float64, ACME_MAX_ELEM_PAR)
    if :βf in names                                   # bjt, src/elements.jl:323-401
        g = (n, default) -> hascaptured(f, n) ? Float64(captured(f, n)) : default
        ηe, ηc = g(:ηe, 1.0), g(:ηc, 1.0)
        par[1:14] = [g(:ise, 1e-12), g(:isc, 1e-12), ηe, ηc, g(:βf, 1000.0), g(:βr, 10.0),
                     g(:ile, 0.0), g(:ilc, 0.0), g(:ηel, ηe), g(:ηcl, ηc),
                     g(:vaf, Inf), g(:var, Inf), g(:ikf, Inf), g(:ikr, Inf)]
        return ACME_KIND_BJT, par
    elseif :is in names && :η in names                # diode, :238-244
        par[1:2] = [Float64(captured(f, :is)), Float64(captured(f, :η))]
        return ACME_KIND_DIODE, par
    elseif :polarity in names && :vt in names         # mosfet, :453-479
        vt, α = captured(f, :vt), captured(f, :α)
        (length(vt) <= 4 && length(α) <= 4) || error("mosfet: at most 4 polynomial coefficients are supported")
        par[1] = Float64(captured(f, :polarity)); par[2] = Float64(captured(f, :λ))
        par[3] = length(vt); par[4:3+length(vt)] .= Float64.(vt)
        par[8] = length(α); par[9:8+length(α)] .= Float64.(α)
        return ACME_KIND_MOSFET, par
    elseif :gain in names && :scale in names          # opamp(Val{:macak}, ...), :540-546
        par[1:2] = [Float64(captured(f, :gain)), Float64(captured(f, :scale))]
        return ACME_KIND_MACAK, par
    elseif :Ms in names                               # Jiles-Atherton core, :107-129
        par[1:5] = Float64[captured(f, :Ms), captured(f, :a), captured(f, :α), captured(f, :c), captured(f, :k)]
        return ACME_KIND_JA, par
    elseif names == (:r,)                             # potentiometer(r), :25-30
        par[1] = Float64(captured(f, :r))
        return ACME_KIND_POT, par
    end
    error("ACMEHip: nonlinear element of unknown kind (captures $(names)); the device code knows " *
          "diode, bjt, potentiometer, mosfet, opamp(Val{:macak}) and the Jiles-Atherton core")
end

"""
    element_table(circ_nl_func) -> (kinds, qoff, roff, par)

The element table of one nonlinear sub-problem, in `CircuitNLFunc` order (src/circuit.jl:68-86).
`circ_nl_func.fs[k]` captures `q_indices` (the element's columns of q) and `nleqfunc`.
"""
function element_table(cnl)
    kinds, qoff, roff = Cint[], Cint[], Cint[]
    par = zeros(Float64, ACME_MAX_ELEM_PAR, length(cnl.fs))   # column-major 16 x n == row-major n x 16
    r = 0
    for (k, f) in enumerate(cnl.fs)
        qi = captured(f, :q_indices)
        kind, p = describe_element(captured(f, :nleqfunc))
        length(qi) == KIND_NQ[Int(kind)] || error("ACMEHip: element $k has $(length(qi)) q rows, kind $kind expects $(KIND_NQ[Int(kind)])")
        push!(kinds, kind); push!(qoff, first(qi) - 1); push!(roff, r)
        par[:, k] = p
        r += KIND_NN[Int(kind)]
    end
    return kinds, qoff, roff, par
end

"`CircuitNLFunc` of sub-problem idx of a model: captured by the closure of src/ACME.jl:176-189"
circuit_nl_func(model::DiscreteModel, idx) = captured(model.nonlinear_eq_funcs[idx], :circ_nl_func)

# (p, z) the innermost solver extrapolates from; HomotopySolver has no get_extrapolation_origin of
# its own in the reference (src/solvers.jl:198,398), so unwrap `.basesolver`
base_origin(s::ACME.SimpleSolver) = get_extrapolation_origin(s)
base_origin(s::NonlinearSolver) = hasfield(typeof(s), :basesolver) ? base_origin(s.basesolver) : get_extrapolation_origin(s)

solver_id(::Type{<:ACME.SimpleSolver}) = ACME_SOLVER_SIMPLE
solver_id(::Type{<:ACME.HomotopySolver{<:ACME.SimpleSolver}}) = ACME_SOLVER_HOMOTOPY
solver_id(::Type{<:ACME.HomotopySolver{<:ACME.CachingSolver}}) = ACME_SOLVER_CACHING_HOMOTOPY   # bounded store, see acme_hip.h
solver_id(::Type{T}) where {T} = error("ACMEHip: no device counterpart of solver type $T")

# ---- acme_model ---------------------------------------------------------------------------------
mutable struct ModelHandle
    h::Ptr{Cvoid}
    function ModelHandle(h)
        m = new(h)
        finalizer(m -> ccall((:acme_model_destroy, lib), Cvoid, (Ptr{Cvoid},), m.h), m)
        return m
    end
end

function add_subproblem!(h::Ptr{Cvoid}, nn, nq, np, pexp, dq, eq, fqprev, fq, q0, init_z, cnl)
    kinds, qoff, roff, par = element_table(cnl)
    check(ccall((:acme_model_add_subproblem, lib), Cint,
        (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}),
        h, nn, nq, np, Matrix{Float64}(pexp), Matrix{Float64}(dq), Matrix{Float64}(eq),
        Matrix{Float64}(fqprev), Matrix{Float64}(fq), Vector{Float64}(q0), Vector{Float64}(init_z),
        length(kinds), kinds, qoff, roff, par))
end

"acme_model of a whole DiscreteModel (all matrices are Matrix{Float64}, column-major: passed as they are)"
function ModelHandle(model::DiscreteModel)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:acme_model_create, lib), Cint,
        (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
        ACME.nx(model), ACME.nu(model), ACME.ny(model), ACME.nn(model),
        model.a, model.b, model.c, model.x0, model.dy, model.ey, model.fy, model.y0, h))
    mh = ModelHandle(h[])
    for idx in 1:length(model.solvers)
        _, init_z = base_origin(model.solvers[idx])                  # (0, init_z) on a fresh model
        add_subproblem!(mh.h, ACME.nn(model, idx), ACME.nq(model, idx), ACME.np(model, idx),
                        model.pexps[idx], model.dqs[idx], model.eqs[idx], model.fqprevs[idx],
                        model.fqs[idx], model.q0s[idx], init_z, circuit_nl_func(model, idx))
    end
    return mh
end

# ---- BatchRunner: N instances of a model on one GPU ------------------------------------------------
"""
    BatchRunner(model::DiscreteModel, n; device=-1, solver=<the model's own solver type>)

N parallel copies of `model` on one MI355X: the batch analogue of `ACME.ModelRunner`
(src/ACME.jl:570-604).  Every instance starts like a fresh model (x = 0, origin (0, init_z)).
"""
mutable struct BatchRunner
    model::DiscreteModel
    n::Int
    h::Ptr{Cvoid}
    mh::ModelHandle
    warned::Int
    progress::Base.RefValue{Any}      # the ProgressMeter.Progress of the run in flight (showprogress = true)
    showprogress::Bool
end

# @showprogress of run!(runner, y, u) (src/ACME.jl:587-604,653): the library reports after every time slice of a
# host-buffer run through a C callback; `user` points at the runner's `progress` cell (kept alive by the runner)
function progress_trampoline(user::Ptr{Cvoid}, done::Clonglong, total::Clonglong)::Cvoid
    p = unsafe_pointer_to_objref(user)::Base.RefValue{Any}
    p[] === nothing || ProgressMeter.update!(p[], Int(done))
    return nothing
end

function BatchRunner(model::DiscreteModel, n::Integer; device::Integer=-1,
                     solver=isempty(model.solvers) ? ACME_SOLVER_HOMOTOPY : solver_id(typeof(model.solvers[1])),
                     per_instance_matrices::Bool=false, showprogress::Bool=false)
    mh = ModelHandle(model)
    opts = Ref(AcmeOptions(solver, 1e-10, 500, device, per_instance_matrices ? 1 : 0))
    b = Ref{Ptr{Cvoid}}()
    check(ccall((:acme_batch_create, lib), Cint, (Ptr{Cvoid}, Clonglong, Ref{AcmeOptions}, Ref{Ptr{Cvoid}}),
                mh.h, n, opts, b))
    r = BatchRunner(model, n, b[], mh, 0, Ref{Any}(nothing), showprogress)
    finalizer(r -> ccall((:acme_batch_destroy, lib), Cvoid, (Ptr{Cvoid},), r.h), r)
    if showprogress
        cb = @cfunction(progress_trampoline, Cvoid, (Ptr{Cvoid}, Clonglong, Clonglong))
        check(ccall((:acme_batch_set_progress_callback, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    r.h, cb, pointer_from_objref(r.progress)))
    end
    return r
end

"""
    retain_host_buffers!(runner, keep=true)

By default `run!` leaves nothing of `u` / `y` registered with the GPU runtime once it returns (they are ordinary,
garbage-collected arrays: a registration must never outlive them).  `retain_host_buffers!(runner)` is the caller's
promise to keep the arrays it passes alive -- reuse them across `run!(runner, y, u)` calls, hold references -- until it
passes others, calls `release_host_buffers!` or drops the runner: they are then page-locked once and runs are streamed
at the device-resident rate (`acme_batch_set_host_retention`, `include/acme_hip.h`).
"""
retain_host_buffers!(r::BatchRunner, keep::Bool=true) =
    (check(ccall((:acme_batch_set_host_retention, lib), Cint, (Ptr{Cvoid}, Cint), r.h, keep ? 1 : 0)); r)

"""
    release_host_buffers!(runner)

Un-page-lock what a retaining runner (`retain_host_buffers!`) holds: before freeing or resizing the arrays while the
runner lives on.
"""
release_host_buffers!(r::BatchRunner) =
    (check(ccall((:acme_batch_release_host_buffers, lib), Cint, (Ptr{Cvoid},), r.h)); r)

"""
    set_isolation!(runner, iters_per_sample)

Launch the instances that needed more than `iters_per_sample` Newton iterations per sample over the previous run on
their own (a pathological cell of a sweep otherwise holds every other instance's results back by its own, hundreds of
times longer, run; `include/acme_hip.h`).  `0` switches it off.
"""
set_isolation!(r::BatchRunner, iters_per_sample::Real) =
    (check(ccall((:acme_batch_set_isolation, lib), Cint, (Ptr{Cvoid}, Cdouble), r.h, iters_per_sample)); r)

"""
    set_balance!(runner, mode)

Placement of a launch's waves by their measured cost (`acme_batch_set_balance`): `-1` lets the library decide (the
default: on when the launch has more blocks than the device has compute units), `0` switches it off, `1` on.
What an instance computes does not depend on it.
"""
set_balance!(r::BatchRunner, mode::Integer) =
    (check(ccall((:acme_batch_set_balance, lib), Cint, (Ptr{Cvoid}, Cint), r.h, mode)); r)

"""
    set_models!(runner, first, models)

Monte-Carlo component tolerances: instances `first`, `first+1`, ... (0-based) take the matrices of
`models[k]` (same circuit topology; the runner must have been created with
`per_instance_matrices=true`) and the freshly constructed state of that model.
"""
function set_models!(r::BatchRunner, first::Integer, models::AbstractVector{<:DiscreteModel})
    hs = [ModelHandle(m) for m in models]
    ptrs = Ptr{Cvoid}[m.h for m in hs]
    GC.@preserve hs check(ccall((:acme_batch_set_matrices, lib), Cint,
        (Ptr{Cvoid}, Clonglong, Clonglong, Ptr{Ptr{Cvoid}}), r.h, first, length(ptrs), ptrs))
    return r
end

function reports(r::BatchRunner)
    reps = Vector{AcmeReport}(undef, r.n)
    check(ccall((:acme_batch_get_report, lib), Cint, (Ptr{Cvoid}, Ptr{AcmeReport}), r.h, reps))
    return reps
end

"""
    run!(runner::BatchRunner, y::Array{Float64,3}, u::Array{Float64,3})
    run!(runner::BatchRunner, u::Array{Float64,3}) -> y

`u` is `nu × T × N`, `y` is `ny × T × N` (Julia's column-major layout of these IS the ABI's
[N][T][nu] layout): slice `[:, :, i]` is exactly the matrix `ACME.run!` takes for instance `i`.
Errors and warnings follow `step!` (src/ACME.jl:688-694) and `checkiosizes` (:625-635).
"""
function checkiosizes(m::DiscreteModel, n::Integer, y::Array{Float64,3}, u::Array{Float64,3})      # src/ACME.jl:625-635
    size(u, 1) == ACME.nu(m) || throw(DimensionMismatch("input matrix has $(size(u,1)) rows, but model has $(ACME.nu(m)) inputs"))
    size(y, 1) == ACME.ny(m) || throw(DimensionMismatch("output matrix has $(size(y,1)) rows, but model has $(ACME.ny(m)) outputs"))
    size(u, 2) == size(y, 2) || throw(DimensionMismatch("input matrix has $(size(u,2)) columns, output matrix has $(size(y,2)) columns"))
    (size(u, 3) == n && size(y, 3) == n) || throw(DimensionMismatch("u and y need one nu × T (ny × T) slice per instance ($n)"))
end

"the policy of `step!` (src/ACME.jl:688-694) on the reports a run left behind; `offset`: instances before this batch"
function checkreports!(r::BatchRunner, offset::Integer=0)
    reps = reports(r)
    bad = findfirst(rep -> rep.first_nonfinite >= 0, reps)
    bad === nothing || error("Failed to converge while solving non-linear equation, got non-finite result. " *
                             "(instance $(offset + bad), sample $(reps[bad].first_nonfinite + 1))")
    nwarn = sum(rep -> rep.n_warn, reps)
    if nwarn > r.warned
        @warn "Failed to converge while solving non-linear equation."
        r.warned = nwarn
    end
    return nothing
end

function run!(r::BatchRunner, y::Array{Float64,3}, u::Array{Float64,3})
    checkiosizes(r.model, r.n, y, u)
    r.showprogress && (r.progress[] = ProgressMeter.Progress(size(u, 2)))     # (as @showprogress for n = 1:size(u, 2))
    GC.@preserve r check(ccall((:acme_batch_run, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Clonglong, Cint, Ptr{Cvoid}),
                r.h, u, y, size(u, 2), ACME_MEM_HOST, C_NULL))
    r.showprogress && (ProgressMeter.finish!(r.progress[]); r.progress[] = nothing)
    return checkreports!(r)
end

function run!(r::BatchRunner, u::Array{Float64,3})
    y = Array{Float64,3}(undef, ACME.ny(r.model), size(u, 2), r.n)
    run!(r, y, u)
    return y
end

"""
    run!(r::BatchRunner, y, u_var, u_const, const_rows)

`run!` with CONSTANT input rows: the rows named in `const_rows` (1-based) keep the value `u_const[k, i]` (`u_const`:
nu x N) for the whole call in instance `i` -- a potentiometer position, a supply voltage -- and `u_var`
(nu_var x T x N) holds the other rows in row order.  Where the reference copies column `n` of a full `u` into `ucur`
sample by sample (src/ACME.jl:672-674), the library puts the full rows together on the device: a sweep over three pot
positions of a four-input model moves a quarter of the bytes over the bus.  Results: those of `run!(r, y, u)` on the
materialised `u`, bit for bit.
"""
function run!(r::BatchRunner, y::Array{Float64,3}, u_var::Array{Float64,3}, u_const::Matrix{Float64}, const_rows)
    nu = ACME.nu(r.model)
    mask = UInt64(0)
    for k in const_rows
        1 <= k <= nu || throw(DimensionMismatch("constant row $k of a model with $nu inputs"))
        mask |= UInt64(1) << (k - 1)
    end
    nuv = nu - count_ones(mask)
    size(u_var, 1) == nuv || throw(DimensionMismatch("u_var has $(size(u_var, 1)) rows, $nuv inputs vary"))
    size(u_const) == (nu, r.n) || throw(DimensionMismatch("u_const must be $nu x $(r.n)"))
    size(u_var, 3) == r.n && size(y, 3) == r.n || throw(DimensionMismatch("u_var and y must hold $(r.n) instances"))
    size(y, 1) == ACME.ny(r.model) && size(y, 2) == size(u_var, 2) ||
        throw(DimensionMismatch("output matrix must be $(ACME.ny(r.model)) x $(size(u_var, 2)) x $(r.n)"))
    GC.@preserve r check(ccall((:acme_batch_run_const, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Culonglong, Ptr{Cdouble}, Clonglong, Cint, Ptr{Cvoid}),
                r.h, u_var, u_const, mask, y, size(u_var, 2), ACME_MEM_HOST, C_NULL))
    return checkreports!(r)
end

# ---- MultiBatchRunner: N instances over the GPUs of one node, one Julia process -----------------------
"contiguous instance range (1-based) of part `k` of `parts`; sizes differ by at most one (acme_jl_amd/dist.py)"
function shard_range(n::Integer, k::Integer, parts::Integer)
    base, rem = divrem(n, parts)
    lo = (k - 1) * base + min(k - 1, rem)
    return lo+1:lo+base+(k <= rem ? 1 : 0)
end

"""
    MultiBatchRunner(model, n; devices=0:device_count()-1, solver=..., per_instance_matrices=false)

`n` instances of `model` spread over several GPUs by ONE process: contiguous instance ranges, one
`BatchRunner` per device.  `run!` starts every device's run asynchronously (`acme_batch_run_async`: the
library drives each from a worker thread of its own) and then joins them (`acme_batch_wait`); every
batch reads and writes its own `[:, :, range]` slice of `u` / `y` in place.  The sweep shards perfectly
(instances never interact), so there is no collective and no MPI/RCCL dependency on the Julia side.
"""
struct MultiBatchRunner
    model::DiscreteModel
    n::Int
    runners::Vector{BatchRunner}
    ranges::Vector{UnitRange{Int}}
end

function MultiBatchRunner(model::DiscreteModel, n::Integer; devices=0:device_count()-1, kwargs...)
    isempty(devices) && error("libacme_hip: no HIP device available")
    ranges = [shard_range(n, k, length(devices)) for k in 1:length(devices)]
    keep = [k for k in 1:length(devices) if !isempty(ranges[k])]
    runners = [BatchRunner(model, length(ranges[k]); device=devices[k], kwargs...) for k in keep]
    return MultiBatchRunner(model, n, runners, ranges[keep])
end

function run!(mr::MultiBatchRunner, y::Array{Float64,3}, u::Array{Float64,3})
    checkiosizes(mr.model, mr.n, y, u)
    T = size(u, 2)
    su, sy = size(u, 1) * T, size(y, 1) * T                   # doubles per instance
    GC.@preserve u y begin
        started = 0
        rcs, msg = Cint[], ""                                 # (assigned in the finally clause below)
        try
            for (r, rg) in zip(mr.runners, mr.ranges)
                check(ccall((:acme_batch_run_async, lib), Cint,
                            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Clonglong, Cint, Ptr{Cvoid}),
                            r.h, pointer(u) + 8 * su * (first(rg) - 1), pointer(y) + 8 * sy * (first(rg) - 1),
                            T, ACME_MEM_HOST, C_NULL))
                started += 1
            end
        finally                                               # never leave a run in flight behind
            rcs = [ccall((:acme_batch_wait, lib), Cint, (Ptr{Cvoid},), r.h) for r in mr.runners[1:started]]
            msg = lasterror()
        end
        all(rc -> rc >= 0, rcs) || error("libacme_hip: " * msg)
    end
    for (r, rg) in zip(mr.runners, mr.ranges)
        checkreports!(r, first(rg) - 1)
    end
    return nothing
end

function run!(mr::MultiBatchRunner, u::Array{Float64,3})
    y = Array{Float64,3}(undef, ACME.ny(mr.model), size(u, 2), mr.n)
    run!(mr, y, u)
    return y
end

reports(mr::MultiBatchRunner) = reduce(vcat, [reports(r) for r in mr.runners])
set_resabstol!(mr::MultiBatchRunner, tol) = (foreach(r -> set_resabstol!(r, tol), mr.runners); tol)

set_resabstol!(r::BatchRunner, tol) =
    (check(ccall((:acme_batch_set_resabstol, lib), Cint, (Ptr{Cvoid}, Cdouble), r.h, tol)); tol)

"(x, last_p, last_z) of all instances: nx × N, Σnp × N, Σnn × N"
function get_state(r::BatchRunner)
    m = r.model
    x = zeros(ACME.nx(m), r.n)
    p = zeros(sum(ACME.np(m, k) for k in 1:length(m.solvers); init=0), r.n)
    z = zeros(ACME.nn(m), r.n)
    check(ccall((:acme_batch_get_state, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), r.h, x, p, z))
    return x, p, z
end

# ---- GPUBatchSolver: the solver plugin contract (src/solvers.jl:139-205) ---------------------------
"""
    GPUBatchSolver(nleq::ParametricNonLinEq, initial_p, initial_z)

A `NonlinearSolver` whose `solve` runs on the GPU (HomotopySolver{SimpleSolver} semantics: Newton
from the first-order extrapolated start, bisection homotopy on failure).  Built, like the
reference's solvers, from the equation object alone (src/ACME.jl:253-259): the data of the closures
in `nleq` -- `fq` and the `CircuitNLFunc` inside `nleq.func` (src/ACME.jl:176-194), `pexp`/`q0`
inside `nleq.set_p` (:236-244), or the identity parametrisation p == q of the three-argument
`ParametricNonLinEq` constructor (src/solvers.jl:23-28) -- is packed as a one-sub-problem model
without states.  Usable wherever ACME takes a solver type:

    model = DiscreteModel(circ, 1//44100, ACMEHip.GPUBatchSolver)
"""
mutable struct GPUBatchSolver <: NonlinearSolver
    mh::ModelHandle
    h::Ptr{Cvoid}
    nn::Int
    np::Int
    z::Vector{Float64}
    converged::Bool
    iters::Int
    function GPUBatchSolver(nleq::ParametricNonLinEq, initial_p::Vector{Float64}, initial_z::Vector{Float64})
        nn_, np_ = ACME.nn(nleq), ACME.np(nleq)
        f = nleq.func                              # (res, J, scratch, z) -> nleq´(res, J, scratch[1], scratch[2], fq, z)
        fq = Matrix{Float64}(captured(f, :fq))
        cnl = captured(captured(f, :nleq), :circ_nl_func)
        nq_ = size(fq, 1)
        if hascaptured(nleq.set_p, :pexp)          # src/ACME.jl:236-244
            pexp = Matrix{Float64}(captured(nleq.set_p, :pexp))
            q0 = Vector{Float64}(captured(nleq.set_p, :q0))
        else                                       # default_set_p: p == q
            np_ == nq_ || error("ACMEHip: identity parametrisation needs np == nq")
            pexp = Matrix{Float64}(I, nq_, nq_)
            q0 = zeros(nq_)
        end
        h = Ref{Ptr{Cvoid}}()
        e = Float64[]
        check(ccall((:acme_model_create, lib), Cint,
            (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
             Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
            0, 0, 0, nn_, e, e, e, e, e, e, e, e, h))
        mh = ModelHandle(h[])
        add_subproblem!(mh.h, nn_, nq_, np_, pexp, zeros(np_, 0), zeros(np_, 0), zeros(np_, nn_), fq, q0,
                        initial_z, cnl)
        opts = Ref(AcmeOptions(ACME_SOLVER_HOMOTOPY, 1e-10, 500, -1, 0))
        b = Ref{Ptr{Cvoid}}()
        check(ccall((:acme_batch_create, lib), Cint, (Ptr{Cvoid}, Clonglong, Ref{AcmeOptions}, Ref{Ptr{Cvoid}}),
                    mh.h, 1, opts, b))
        s = new(mh, b[], nn_, np_, copy(initial_z), true, 0)
        finalizer(s -> ccall((:acme_batch_destroy, lib), Cvoid, (Ptr{Cvoid},), s.h), s)
        set_extrapolation_origin(s, initial_p, initial_z)
        return s
    end
end

"z = solve(solver, p): returns the solver's internal buffer, like the reference (the caller copies, src/ACME.jl:695)"
function solve(s::GPUBatchSolver, p::AbstractVector{Float64})
    conv, iters = Ref{Cint}(0), Ref{Cint}(0)
    check(ccall((:acme_batch_solve, lib), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}, Cint, Ptr{Cvoid}),
                s.h, 0, Vector{Float64}(p), s.z, conv, iters, ACME_MEM_HOST, C_NULL))
    s.converged = conv[] != 0
    s.iters = iters[]
    return s.z
end

hasconverged(s::GPUBatchSolver) = s.converged
needediterations(s::GPUBatchSolver) = s.iters
set_resabstol!(s::GPUBatchSolver, tol) =
    (check(ccall((:acme_batch_set_resabstol, lib), Cint, (Ptr{Cvoid}, Cdouble), s.h, tol)); tol)

function get_extrapolation_origin(s::GPUBatchSolver)
    p, z = zeros(s.np), zeros(s.nn)
    check(ccall((:acme_batch_get_state, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                s.h, C_NULL, p, z))
    return p, z
end

function set_extrapolation_origin(s::GPUBatchSolver, p, z)
    check(ccall((:acme_batch_set_state, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                s.h, C_NULL, Vector{Float64}(p), Vector{Float64}(z)))
    return nothing
end

"-(J \\ Jp) at the extrapolation origin (src/solvers.jl:198-201), nn × np"
function get_extrapolation_jacobian(s::GPUBatchSolver)
    jac = zeros(s.nn, s.np)
    check(ccall((:acme_batch_get_extrapolation_jacobian, lib), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint, Ptr{Cvoid}), s.h, 0, jac, ACME_MEM_HOST, C_NULL))
    return jac
end

end # module
