# patch_neural_ode.jl -- the four call methods of TrackedNeuralODE (reference src/models/neural_ode.jl:48-180) with their `solve` replaced
# by librnde.so.  SOURCE ONLY (no Julia in the build image).  Usage, in an experiment script such as experiments/mnist_node.jl:
#
#     using RegNeuralDE
#     include("/path/to/repo/bindings/julia/RNDE.jl")
#     include("/path/to/repo/bindings/julia/patch_neural_ode.jl")      # redefines the methods below; everything else is untouched
#
# Signatures, keyword defaults, `_convert_tspan` (src/utils.jl:21-23), the returned triple `(res, nfe, sv)` and the `SavedValues` container are
# the reference's.  `func` is the caller's closure, exactly as the unchanged experiment passes it (`model(x, p1, p2, p3; func = save_func, ...)`,
# mnist_node.jl:134): RNDE.reg_code evaluates it on two mock integrators and recognises which of the reference's callbacks it is
# (mnist_node.jl:67 / :74-79 / :88-97), the handle is created with that `regularize` code and the library records EEst * dt, the stiffness
# estimate or their blend inside its kernels.  An unrecognised closure is an error, never a silently different loss.
#
# The layer struct has no field for the handle (and is immutable), so handles live in a table keyed by the layer object, the batch width and
# the callback code.
using Tracker, Flux, DiffEqCallbacks
using RegNeuralDE: TrackedNeuralODE, TDChain, _convert_tspan

const RNDE_ODE_HANDLES = IdDict{Any,Dict{Tuple{Int,Int},RNDE.Handle}}()
const TSIT5_STABILITY_SIZE = 3.5068      # OrdinaryDiffEq.alg_stability_size(Tsit5()), the constant mnist_node.jl:73,:86 divides by

# Dense sizes / activations of the dynamics (TDChain or Chain of Dense layers; a leading `x -> tanh.(x)` is latent_ode.jl:114's pre-activation).
# Anything the library cannot represent is REFUSED here -- a layer that is silently skipped would integrate another vector field:
# every layer must be a Flux.Dense with tanh or identity, except ONE leading element-wise function that is tanh (checked on a probe vector).
_act_code(σ) = σ === tanh ? 1 : (σ === identity ? 0 : error("RNDE: Dense activation ", σ, " is not served (tanh / identity)"))
_is_tanh_layer(l) = !(l isa Flux.Dense) && (v = Float32[-0.7, 0.1, 0.9]; try l(v) ≈ tanh.(v) catch; false end)
function _dense_layout(model)
    layers = collect(model.layers)
    td = model isa TDChain
    pre = !(first(layers) isa Flux.Dense)
    pre && (!td && _is_tanh_layer(first(layers)) ||
            error("RNDE: the dynamics may start with ONE element-wise tanh (experiments/latent_ode.jl:114), nothing else in front of the Dense layers; got ", first(layers)))
    ds = layers[(pre ? 2 : 1):end]
    all(l -> l isa Flux.Dense, ds) || error("RNDE: the dynamics must be a chain of Flux.Dense layers; got ", [typeof(l) for l in ds if !(l isa Flux.Dense)])
    dims = Int[size(ds[1].W, 2) - (td ? 1 : 0)]
    acts = Int[]
    for l in ds
        push!(dims, size(l.W, 1)); push!(acts, _act_code(l.σ))
    end
    return dims, acts, td, pre
end

# regularize code of include/rnde.h from the type parameter R, the caller's `func` and the solver the layer holds (n.args)
function _reg_code(n::TrackedNeuralODE{R}, func) where {R}
    name, composite = RNDE.solver_name(n.args)
    name === :Tsit5 || error("RNDE: the ODE layer runs Tsit5() / AutoTsit5(Tsit5()) (every reference call site); got ", name)
    R || return RNDE.REG_NONE
    return RNDE.effective_reg(RNDE.reg_code(func, TSIT5_STABILITY_SIZE), composite)
end

function rnde_handle(n::TrackedNeuralODE, B::Int, code::Int)
    tab = get!(() -> Dict{Tuple{Int,Int},RNDE.Handle}(), RNDE_ODE_HANDLES, n)
    get!(tab, (B, code)) do
        dims, acts, td, pre = _dense_layout(n.model)
        RNDE.Handle(RNDE.config_for(dims, acts; time_dep = td, pre_act = pre, max_batch = B, reltol = Float32(get(n.kwargs, :reltol, 1f-3)),
                                    abstol = Float32(get(n.kwargs, :abstol, 1f-6)), regularize = code))
    end
end

_saveat_vec(n, saveat) = Float32.(collect(Tracker.data(isnothing(saveat) ? n.kwargs[:saveat] : saveat)))      # (a tracked / device vector of save times is read back: they are host data for the library)
# which times a multi-output call saves at: the given / stored saveat -- or, for a layer built with save_everystep = true and no saveat
# (neural_ode.jl:10-11), the ends of the accepted steps, which an untracked solve has to find first (RNDE.solve_forward_everystep); the tracked
# call then saves at exactly those (the value at a step's end is u_new itself)
function _multi_times(n, h, x, p, tspan, saveat)
    (!isnothing(saveat) || haskey(n.kwargs, :saveat)) && return _saveat_vec(n, saveat)
    _, ts, _, _ = RNDE.solve_forward_everystep(h, Tracker.data(x), Tracker.data(p), tspan;
                                               save_start = get(n.kwargs, :save_start, true), keep_tape = false)
    return ts
end
_saved(tspan, p, saveval) = (sv = SavedValues(eltype(tspan), eltype(p)); append!(sv.saveval, saveval); sv)

# {false,false} (reference :48-77): vanilla solve, end state only
function (n::TrackedNeuralODE{false,false})(x, p = n.p; func = (u, t, int) -> 0, tspan = nothing, saveat = nothing)
    tspan = _convert_tspan(isnothing(tspan) ? n.tspan : tspan, p)
    h = rnde_handle(n, size(x, 2), _reg_code(n, func))
    res, _ = RNDE.rnde_solve(h, x, p, tspan)                                   # <- replaces :61-70
    return res, h.last_nfe, nothing
end

# {false,true} (reference :79-108): all saved states, D x T x B
function (n::TrackedNeuralODE{false,true})(x, p = n.p; func = (u, t, int) -> 0, tspan = nothing, saveat = nothing)
    tspan = _convert_tspan(isnothing(tspan) ? n.tspan : tspan, p)
    h = rnde_handle(n, size(x, 2), _reg_code(n, func))
    res, _ = RNDE.rnde_solve_saveat(h, x, p, tspan, _multi_times(n, h, x, p, tspan, saveat))     # <- replaces :92-102 (update_saveat! is not needed: nothing is mutated)
    return res, h.last_nfe, nothing
end

# {true,false} (reference :110-144): end state + the saving callback's values (configs 2 / 3)
function (n::TrackedNeuralODE{true,false})(x, p = n.p; func = (u, t, integrator) -> integrator.EEst * integrator.dt, tspan = nothing,
                                           saveat = nothing)
    tspan = _convert_tspan(isnothing(tspan) ? n.tspan : tspan, p)
    h = rnde_handle(n, size(x, 2), _reg_code(n, func))
    res, saveval = RNDE.rnde_solve(h, x, p, tspan)                             # <- replaces :126-138
    return res, h.last_nfe, _saved(tspan, p, saveval)
end

# {true,true} (reference :146-180): saved states + callback values (config 4, latent_ode.jl:137-147)
function (n::TrackedNeuralODE{true,true})(x, p = n.p; func = (u, t, integrator) -> integrator.EEst * integrator.dt, tspan = nothing,
                                          saveat = nothing)
    tspan = _convert_tspan(isnothing(tspan) ? n.tspan : tspan, p)
    h = rnde_handle(n, size(x, 2), _reg_code(n, func))
    res, saveval = RNDE.rnde_solve_saveat(h, x, p, tspan, _multi_times(n, h, x, p, tspan, saveat))   # <- replaces :162-174
    return res, h.last_nfe, _saved(tspan, p, saveval)
end
