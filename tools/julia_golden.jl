# tools/julia_golden.jl -- closes "parity unpinned" on a host that HAS Julia and the reference's environment.
#
# Runs the fixture inputs of tests/golden/make_golden.py (same portable LCG, same shapes, same tolerances) through the REAL
# reference layers -- TrackedNeuralODE (src/models/neural_ode.jl:48-77, :110-144) with Tsit5, and TrackedNeuralDSDE
# (src/models/neural_sde.jl:116-146) with SOSRI -- and dumps, per case, what the oracle and the device are compared on:
#     u_end, nfe, saveval, and the gradients of  sum(wu .* u_end) + 25 * sum(saveval)  with respect to x and p.
# Output: one CSV-ish text file per case under tests/golden/julia/ (plain text so that no Julia package beyond the
# reference's own Manifest is needed).  tests/test_oracle.py::test_julia_golden_if_present picks them up and compares the
# fp32 / fp64 oracle against them (skipped while the directory is absent: this container has no Julia).
#
#   cd /path/to/RegNeuralDE.jl && julia --project=. /path/to/repo/tools/julia_golden.jl /path/to/repo/tests/golden/julia
#
# Run it on the CPU (CUDA_VISIBLE_DEVICES="") unless the host has the reference's CUDA setup; results are fp32 either way.
# Call shapes follow the reference's own test (test/test_node.jl:4-57).
using RegNeuralDE, OrdinaryDiffEq, StochasticDiffEq, Flux, Tracker, Random, Printf

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(pwd(), "julia_golden")
mkpath(outdir)

# --- the portable generator of tests/golden/make_golden.py (Knuth MMIX LCG, top 53 bits) -----------------------------------
function lcg_uniform(n::Int, seed::Integer, lo = 0.0, hi = 1.0)
    out = Vector{Float64}(undef, n)
    s = UInt64(seed)
    a, c = UInt64(6364136223846793005), UInt64(1442695040888963407)
    for i = 1:n
        s = s * a + c                      # wraps modulo 2^64
        out[i] = Float64(s >> 11) / Float64(UInt64(1) << 53)
    end
    return lo .+ (hi - lo) .* out
end

# parameters in Flux.destructure order: per layer vec(W) (out x in_ext, column-major) then b
function params_for(dims, time_dep, seed, scale)
    parts = Float64[]
    for l = 1:length(dims)-1
        ine = dims[l] + (time_dep ? 1 : 0)
        o = dims[l+1]
        lim = scale * sqrt(6.0 / (ine + o))
        append!(parts, lcg_uniform(ine * o, seed + 17 * (l - 1), -lim, lim))
        append!(parts, lcg_uniform(o, seed + 17 * (l - 1) + 5, -0.05, 0.05))
    end
    return Float32.(parts)
end

function dump(path, pairs)
    open(path, "w") do io
        for (k, v) in pairs
            vals = v isa Number ? [v] : vec(collect(v))
            println(io, k, " ", length(vals), " ", join((@sprintf("%.9e", Float64(x)) for x in vals), " "))
        end
    end
end

# --- ODE cases (name, model builder, dims, time_dep, B, tol, scale, t1, seed): tests/golden/make_golden.py CASES ---------------
tanh_ = tanh
ode_cases = [
    ("test_node_B1", () -> TDChain(Dense(3, 10, tanh_), Dense(11, 2)), [2, 10, 2], true, 1, 1f-3, 3.0, 1f0, 11),
    ("test_node_B5", () -> TDChain(Dense(3, 10, tanh_), Dense(11, 2)), [2, 10, 2], true, 5, 1f-3, 3.0, 1f0, 12),
    ("mnist_small_B4", () -> TDChain(Dense(37, 10, tanh_), Dense(11, 36, tanh_)), [36, 10, 36], true, 4, 1f-3, 4.0, 1f0, 13),
    ("mnist_B3", () -> TDChain(Dense(785, 100, tanh_), Dense(101, 784, tanh_)), [784, 100, 784], true, 3, 1f-3, 3.0, 1f0, 14),
    # the headline tolerance on the reference test's own shape (test/test_node.jl:10-19): pins the NFE the fp32 noise floor gives Julia
    ("test_node_B5_tol1.4e-8", () -> TDChain(Dense(3, 10, tanh_), Dense(11, 2)), [2, 10, 2], true, 5, 1.4f-8, 3.0, 1f0, 12),
    ("mnist_B64_tol1.4e-8", () -> TDChain(Dense(785, 100, tanh_), Dense(101, 784, tanh_)), [784, 100, 784], true, 64, 1.4f-8, 1.0, 1f0, 21),
]

for (name, mk, dims, td, B, tol, scale, t1, seed) in ode_cases
    model = mk() |> track
    D = dims[1]
    p = params_for(dims, td, seed, scale)
    x = Float32.(reshape(lcg_uniform(B * D, seed + 1000), D, B))              # column-major D x B == the fixtures' (B, D) row-major
    wu = Float32.(reshape(lcg_uniform(B * D, seed + 2000, -1.0, 1.0), D, B))
    node = TrackedNeuralODE(model, [0.0f0, t1], td, true, Tsit5(), save_everystep = false, reltol = tol, abstol = tol, save_start = false)
    @assert length(node.p) == length(p) "Flux.destructure length differs from the documented layout"
    func = (u, t, integrator) -> integrator.EEst * integrator.dt
    res, nfe, sv = node(x |> track, p |> track; func = func)
    loss(xx, pp) = begin
        r, _, s = node(xx, pp; func = func)
        sum(wu .* r) + 25 * sum(s.saveval)
    end
    gx, gp = Tracker.gradient(loss, x, p)
    dump(joinpath(outdir, name * ".txt"), ["u" => Tracker.data(res), "nfe" => nfe, "saveval" => Tracker.data.(sv.saveval),
                                           "xbar" => Tracker.data(gx), "pbar" => Tracker.data(gp), "tol" => tol, "B" => B])
    println(name, ": nfe ", nfe, ", |saveval| ", length(sv.saveval))
end

# --- ODE cases with the STIFFNESS callback (experiments/mnist_node.jl:70-83: AutoTsit5(Tsit5()), |eigen_est| / alg_stability_size(Tsit5()), agg = maximum):
# pins the [RECALL] formula eigen_est = rms(k7 - k6) / rms(u - g6) and its reverse.  Same inputs as the cases above; file NAME_stiff.txt.
let stability_size = 1 / Float32(OrdinaryDiffEq.alg_stability_size(Tsit5()))
    save_func(u, t, integrator) = (s = abs(integrator.eigen_est); stability_size * ((iszero(s) || isnan(s)) ? 0 : s))
    for (name, mk, dims, td, B, tol, scale, t1, seed) in ode_cases[1:4]
        model = mk() |> track
        D = dims[1]
        p = params_for(dims, td, seed, scale)
        x = Float32.(reshape(lcg_uniform(B * D, seed + 1000), D, B))
        wu = Float32.(reshape(lcg_uniform(B * D, seed + 2000, -1.0, 1.0), D, B))
        node = TrackedNeuralODE(model, [0.0f0, t1], td, true, AutoTsit5(Tsit5()), save_everystep = false, reltol = tol, abstol = tol, save_start = false)
        res, nfe, sv = node(x |> track, p |> track; func = save_func)
        loss(xx, pp) = begin
            r, _, s = node(xx, pp; func = save_func)
            sum(wu .* r) + 25 * sum(s.saveval)
        end
        gx, gp = Tracker.gradient(loss, x, p)
        dump(joinpath(outdir, name * "_stiff.txt"), ["u" => Tracker.data(res), "nfe" => nfe, "saveval" => Tracker.data.(sv.saveval),
                                                     "xbar" => Tracker.data(gx), "pbar" => Tracker.data(gp), "tol" => tol, "B" => B,
                                                     "stability_size" => OrdinaryDiffEq.alg_stability_size(Tsit5())])
        println(name, "_stiff: nfe ", nfe, ", saveval[1:3] ", Tracker.data.(sv.saveval)[1:min(3, end)])
    end
end

# --- SDE case: config-5 shapes at B = 24, reltol = abstol = 0.14 (experiments/mnist_nsde.jl:72-84).  Julia's RNG stream cannot be
# fed to the oracle after the fact, so this case pins STATISTICS only (attempt count, nfe1 = nfe2 = 2 + 4 attempts, saveval scale);
# a bitwise comparison would need the noise pool interface of include/rnde.h (rnde_nsde_forward(noise_dev = ...)) bound in Julia.
let B = 24, seed = 31
    drift = Chain(Dense(32, 64, tanh), Dense(64, 32)) |> track
    diff = Dense(32, 32) |> track
    nsde = TrackedNeuralDSDE(drift, diff, [0.0f0, 1.0f0], true, SOSRI(), save_everystep = false, reltol = 1.4f-1, abstol = 1.4f-1, save_start = false)
    p = vcat(params_for([32, 64, 32], false, seed, 2.0), params_for([32, 32], false, seed + 100, 0.5))
    x = Float32.(reshape(lcg_uniform(B * 32, seed + 1000, -1.0, 1.0), 32, B))
    Random.seed!(1999)
    atts = Int[]; svs = Float64[]
    for rep = 1:32
        res, nfe1, nfe2, sv = nsde(x |> track, p |> track)
        @assert nfe1 == nfe2
        push!(atts, (nfe1 - 2) ÷ 4); push!(svs, sum(Tracker.data.(sv.saveval)))
    end
    dump(joinpath(outdir, "nsde_B24_stats.txt"), ["attempts" => atts, "sum_saveval" => svs])
    println("nsde_B24: attempts ", atts)
end

# --- SDE, the SHIPPED default (experiments/configs/mnist_nsde.yml:6 `type: stiff_est` -> mnist_nsde.jl:51-61): AutoSOSRI2(SOSRI2()) and
# |eigen_est| / alg_stability_size(SOSRI2()).  Statistics again (the RNG stream cannot be shared): per run the attempt count, the FIRST saved value
# (the callback's initialisation: pins eigen_est's initial value, [RECALL] 1 -> 1 / 10.6), the mean of the others (pins the scale of
# rms(k4 - k3) / rms(H0_4 - H0_3), [RECALL]) -- and the constant itself.  The oracle / device counterpart: tests/test_gpu_nsde.py::test_stiffness_estimate_*.
let B = 24, seed = 41
    stab = StochasticDiffEq.alg_stability_size(SOSRI2())
    stability_size = 1 / Float32(stab)
    save_func(u, t, integrator) = (s = abs(integrator.eigen_est); stability_size * ((iszero(s) || isnan(s)) ? 0 : s))
    drift = Chain(Dense(32, 64, tanh), Dense(64, 32)) |> track
    diff = Dense(32, 32) |> track
    nsde = TrackedNeuralDSDE(drift, diff, [0.0f0, 1.0f0], true, AutoSOSRI2(SOSRI2()), save_everystep = false, reltol = 1.4f-1, abstol = 1.4f-1, save_start = false)
    p = vcat(params_for([32, 64, 32], false, seed, 2.0), params_for([32, 32], false, seed + 500, 0.5))      # = tests/golden/make_golden.py nsde_stiff_inputs
    x = Float32.(reshape(lcg_uniform(B * 32, seed + 1000, -1.0, 1.0), 32, B))
    Random.seed!(1999)
    atts = Int[]; first_sv = Float64[]; mean_sv = Float64[]
    for rep = 1:32
        res, nfe1, nfe2, sv = nsde(x |> track, p |> track; func = save_func)
        v = Float64.(Tracker.data.(sv.saveval))
        push!(atts, (nfe1 - 2) ÷ 4); push!(first_sv, v[1]); push!(mean_sv, sum(v[2:end]) / max(1, length(v) - 1))
    end
    dump(joinpath(outdir, "nsde_stiff_B24_stats.txt"), ["attempts" => atts, "first_saveval" => first_sv, "mean_saveval_after_first" => mean_sv, "stability_size" => stab])
    println("nsde_stiff_B24: attempts ", atts, " mean saved value ", sum(mean_sv) / length(mean_sv), " stability size ", stab)
end
