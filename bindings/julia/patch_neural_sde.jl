# patch_neural_sde.jl -- the four call methods of TrackedNeuralDSDE (reference src/models/neural_sde.jl:44-146) with their `solve` replaced by
# librnde.so.  SOURCE ONLY (no Julia in the build image).  Usage: include RNDE.jl, then this file, after `using RegNeuralDE` (see
# patch_neural_ode.jl).  Signatures and the returned 4-tuple `(arr, nfe1, nfe2, sv)` are the reference's; the drift / diffusion evaluation
# counts come from the library (2 + 4 per attempted step each: what the closures' counters at :46, :50 count), the layer's own `nfes`
# vector is left at zero as the reference leaves it after every call (:142-143).
using Tracker, Flux, DiffEqCallbacks
using RegNeuralDE: TrackedNeuralDSDE, _convert_tspan

const RNDE_SDE_HANDLES = IdDict{Any,Dict{Tuple{Int,Int},RNDE.NsdeHandle}}()
const SOSRI2_STABILITY_SIZE = 10.6       # StochasticDiffEq.alg_stability_size(SOSRI2()), the constant mnist_nsde.jl:55 divides by
const _SDE_SOLVERS = Dict(:SOSRI => 0, :SRIW1 => 1, :SOSRI2 => 2)      # rnde_sde_solver (include/rnde.h)
const RNDE_SDE_CALLS = Ref(0)      # one Philox stream per call: seed = a counter (pass `seed = ...` for a reproducible run)

# drift / diffusion as the library holds them: chains of Flux.Dense (tanh / identity), nothing else.  A layer that cannot be represented is REFUSED,
# never skipped (experiments/sde_toy_problem.jl's drift starts with `x -> x .^ 3`: that script is not served by this patch and now says so).
function _chain_layout(model)
    ls = model isa Flux.Dense ? [model] : collect(model.layers)
    all(l -> l isa Flux.Dense, ls) || error("RNDE: drift and diffusion of the SDE layer must be chains of Flux.Dense layers; got ", [typeof(l) for l in ls if !(l isa Flux.Dense)])
    dims = Int[size(ls[1].W, 2)]; acts = Int[]
    for l in ls
        push!(dims, size(l.W, 1))
        push!(acts, l.σ === tanh ? 1 : (l.σ === identity ? 0 : error("RNDE: Dense activation ", l.σ, " is not served (tanh / identity)")))
    end
    return dims, acts
end

# solver + regularize codes of include/rnde.h from the solver object the layer holds (n.args: SOSRI() / AutoSOSRI2(SOSRI2()), mnist_nsde.jl:49,:60),
# the type parameter R and the caller's `func` (`model(x, p1, p2, p3, p4; func = save_func)`): recognised by RNDE.reg_code, never called per step
function _sde_codes(n::TrackedNeuralDSDE{R}, func) where {R}
    name, composite = RNDE.solver_name(n.args)
    isempty(n.args) && (name = :SOSRI)
    haskey(_SDE_SOLVERS, name) || error("RNDE: the SDE layer runs SOSRI() / SOSRI2() / AutoSOSRI2(SOSRI2()) / SRIW1(); got ", name)
    reg = R ? RNDE.effective_reg(RNDE.reg_code(func, SOSRI2_STABILITY_SIZE), composite) : RNDE.REG_NONE
    reg in (RNDE.REG_NONE, RNDE.REG_ERR) || name === :SOSRI2 ||
        error("RNDE: the stiffness estimate of an SRI step is defined for SOSRI2 only (its last two drift stages share one time); got ", name)
    (reg == RNDE.REG_ERR_STIFF || reg == RNDE.REG_STIFF_DT) && error("RNDE: the SDE layer records EEst*dt or the stiffness estimate (mnist_nsde.jl:45-61), not their blend and not |eigen_est*dt|")
    return _SDE_SOLVERS[name], reg
end

function rnde_handle(n::TrackedNeuralDSDE, B::Int, func)
    solver, reg = _sde_codes(n, func)
    tab = get!(() -> Dict{Tuple{Int,Int},RNDE.NsdeHandle}(), RNDE_SDE_HANDLES, n)
    get!(tab, (B, reg)) do
        d1, a1 = _chain_layout(n.model1); d2, a2 = _chain_layout(n.model2)
        RNDE.NsdeHandle(RNDE.nsde_config_for(d1, a1, d2, a2; max_batch = B, reltol = Float32(get(n.kwargs, :reltol, 1f-2)),
                                             abstol = Float32(get(n.kwargs, :abstol, 1f-2)), regularize = reg, solver = solver))
    end
end

_sde_saved(tspan, p, saveval) = (sv = SavedValues(eltype(tspan), eltype(p)); append!(sv.saveval, saveval); sv)
_sde_saveat(n) = Float32.(collect(Tracker.data(n.kwargs[:saveat])))
_next_seed(seed) = isnothing(seed) ? (RNDE_SDE_CALLS[] += 1) : seed

# {false,false} (reference :63-82)
function (n::TrackedNeuralDSDE{false,false})(x, p = n.p; func = (u, t, int) -> 0, seed = nothing)
    tspan = _convert_tspan(n.tspan, p)
    h = rnde_handle(n, size(x, 2), func)
    arr, _ = RNDE.rnde_nsde_solve(h, x, p, tspan, _next_seed(seed))                            # <- replaces :74-76
    c = RNDE.counters(h)
    return arr, c.nfe1, c.nfe2, nothing
end

# {false,true} (reference :44-61): all saved states, D x T x B
function (n::TrackedNeuralDSDE{false,true})(x, p = n.p; func = (u, t, int) -> 0, seed = nothing)
    tspan = _convert_tspan(n.tspan, p)
    h = rnde_handle(n, size(x, 2), func)
    arr, _ = RNDE.rnde_nsde_solve_saveat(h, x, p, tspan, _sde_saveat(n), _next_seed(seed))      # <- replaces :54-56
    c = RNDE.counters(h)
    return arr, c.nfe1, c.nfe2, nothing
end

# {true,false} (reference :116-146): end state + the saving callback's value per accepted step -- EEst * dt or |eigen_est| / 10.6, whichever
# `func` is (config 5, experiments/mnist_nsde.jl:45-61; the shipped configs/mnist_nsde.yml selects the stiffness estimate)
function (n::TrackedNeuralDSDE{true,false})(x, p = n.p; func = (u, t, integrator) -> integrator.EEst * integrator.dt, seed = nothing)
    tspan = _convert_tspan(n.tspan, p)
    h = rnde_handle(n, size(x, 2), func)
    arr, saveval = RNDE.rnde_nsde_solve(h, x, p, tspan, _next_seed(seed))                      # <- replaces :130-140
    c = RNDE.counters(h)
    return arr, c.nfe1, c.nfe2, _sde_saved(tspan, p, saveval)
end

# {true,true} (reference :84-113)
function (n::TrackedNeuralDSDE{true,true})(x, p = n.p; func = (u, t, integrator) -> integrator.EEst * integrator.dt, seed = nothing)
    tspan = _convert_tspan(n.tspan, p)
    h = rnde_handle(n, size(x, 2), func)
    arr, saveval = RNDE.rnde_nsde_solve_saveat(h, x, p, tspan, _sde_saveat(n), _next_seed(seed))   # <- replaces :98-108
    c = RNDE.counters(h)
    return arr, c.nfe1, c.nfe2, _sde_saved(tspan, p, saveval)
end
