# bench_reference.jl -- BASELINE.md section 3, steps B0/B1: time the REAL reference on the measurement box,
# and run the doctest through the binding.  Invoked by bench.py when `julia` is on PATH:
#
#     julia julia/bench_reference.jl <repo-root> <seconds-of-signal> <streams>
#
# Prints ONE JSON line.  Needs the ACME package in the active Julia environment (there is no network on
# the measurement box: if `using ACME` fails, the line says so and bench.py records it).
#   reference_run!      : run!(ModelRunner(model, false), y, u) on the variable-pot superover (BASELINE
#                         config 3's model), default solver stack, `streams` grid cells one after another
#                         on this one process after a warm-up run of the same length (JIT + cache learning)
#   binding_doctest_err : |y - doctest| of the diode clipper run through ACMEHip.BatchRunner
#                         (docs/src/gettingstarted.md:106-113), when a GPU and libacme_hip.so are there
root, seconds, streams = ARGS[1], parse(Float64, ARGS[2]), parse(Int, ARGS[3])
json(d) = "{" * join(["\"$k\": " * (v isa AbstractString ? "\"" * replace(v, "\"" => "'") * "\"" : v === nothing ? "null" : string(v)) for (k, v) in d], ", ") * "}"
out = Dict{String,Any}("julia_version" => string(VERSION), "threads" => Threads.nthreads())
try
    @eval using ACME
catch e
    out["error"] = "using ACME failed: " * sprint(showerror, e)
    println(json(out)); exit(0)
end
examples = joinpath(dirname(dirname(pathof(ACME))), "examples")
include(joinpath(examples, "superover.jl"))
include(joinpath(examples, "diodeclipper.jl"))
T = round(Int, 44100 * seconds)
sig = sin.(2π * 1000 / 44100 .* (0:T-1))
model = superover(DiscreteModel)                       # pots as inputs: nu = 4
function cell(k)                                       # the bench grid's cell k: drive x tone x level
    u = zeros(4, T); u[1, :] = sig
    u[2, :] .= (k % 32) / 32; u[3, :] .= ((k ÷ 32) % 16) / 15; u[4, :] .= ((k ÷ 512) % 16) / 15
    return u
end
let total = 0.0
    for s in 0:streams-1
        m = superover(DiscreteModel)
        r = ModelRunner(m, false)
        u = cell(s * (8192 ÷ max(streams, 1)))
        y = zeros(1, T)
        run!(r, y, u)                                  # warm-up: JIT, solution cache
        total += @elapsed run!(r, y, u)
    end
    out["reference_run!"] = streams * T / total        # instance*samples/s of ONE Julia process
    out["reference_streams"] = streams; out["reference_samples"] = T
end
try
    include(joinpath(root, "julia", "ACMEHip.jl"))
    clip = diodeclipper(DiscreteModel)
    n = 44100
    u = reshape(sin.(2π * 1000 / 44100 .* (0:n-1)), 1, n, 1)
    y = Main.ACMEHip.run!(Main.ACMEHip.BatchRunner(clip, 1), Array(u))
    doc = [0.0, 0.0275964, 0.0990996, 0.195777]
    out["binding_doctest_err"] = maximum(abs.(y[1, 1:4, 1] .- doc))
    out["binding_devices"] = Main.ACMEHip.device_count()
catch e
    out["binding_error"] = sprint(showerror, e)
end
println(json(out))
