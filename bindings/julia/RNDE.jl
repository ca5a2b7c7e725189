# RNDE.jl -- Julia binding of librnde.so (include/rnde.h) for RegNeuralDE.jl on MI355X.
#
# SOURCE ONLY: no Julia toolchain exists in the build image, so this file has never been executed (INTEGRATION.md says what
# was checked instead: every prototype below is the one regneuralde.jl_amd/_lib.py binds with ctypes and the GPU tests call;
# tests/test_abi.py parses the two config structs below and compares field order, widths and offsets with a C program compiled
# against include/rnde.h, tests/abi_c/abi_check.c, and with the ctypes mirrors).
# It shows exactly what a maintainer adds to the reference: the body of the TrackedNeuralODE call methods between ODEProblem
# construction and result unpacking (reference src/models/neural_ode.jl:126-142) becomes one `ccall`, the same for
# TrackedNeuralDSDE (src/models/neural_sde.jl:98-113), and the reverse sweeps are registered with `Tracker.@grad` so that
# `Tracker.gradient(...)` (reference experiments/mnist_node.jl:229-232) keeps working unchanged.
#
# Device arrays are AMDGPU.jl `ROCArray`s (the reference's `|> gpu` becomes `|> roc`).  A device pointer crosses the ABI as a
# plain `Ptr{Cvoid}`: `devptr(a)` below reinterprets AMDGPU's typed device pointer, nothing else about the array is touched.
module RNDE

import AMDGPU                                   # (binds the module name too: AMDGPU.stream(), AMDGPU.synchronize() below)
using AMDGPU: ROCArray, ROCVector, ROCMatrix
using Tracker
using Tracker: TrackedArray, data, track, @grad

const LIB = joinpath(@__DIR__, "..", "..", "regneuralde.jl_amd", "lib", "librnde.so")
const MAX_LAYERS = 8

devptr(a::ROCArray) = reinterpret(Ptr{Cvoid}, pointer(a))

# mirrors rnde_node_config (include/rnde.h); field order and widths must match
struct NodeConfig
    n_layers::Int32
    dims::NTuple{9,Int32}
    act::NTuple{8,Int32}
    time_dep::Int32
    pre_act::Int32
    max_batch::Int32
    solver::Int32
    reltol::Float32
    abstol::Float32
    regularize::Int32
    cb_save_start::Int32
    track_ctrl::Int32
    track_initdt::Int32
    max_attempts::Int32
    device::Int32
    col_tile::Int32
    persist::Int32
    wgrad_side_pct::Int32
    stage_generic::Int32
end

# ---- stream ordering contract ---------------------------------------------------------------------------
# Every entry point below that enqueues work takes the HIP stream of the CURRENT Julia task (AMDGPU.jl arrays live on task-local
# streams): the library's kernels are then ordered against the caller's own array operations on that task, and results read back
# from Julia after the call (or after AMDGPU.synchronize() for the asynchronous ones: backward_async!, classifier_grad!,
# momentum_step!, adam_step!, allreduce_sum!) are complete.  The NULL stream would only be ordered against task streams that
# were created blocking, which AMDGPU.jl does not promise.
function _stream()
    try
        return Base.unsafe_convert(Ptr{Cvoid}, AMDGPU.stream().stream)
    catch
        AMDGPU.synchronize()          # (an AMDGPU.jl without that accessor: fall back to the NULL stream behind a full synchronisation)
        return C_NULL
    end
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    cfg::NodeConfig
    last_nfe::Int
    function Handle(cfg::NodeConfig)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:rnde_node_create, LIB), Cint, (Ref{NodeConfig}, Ref{Ptr{Cvoid}}), cfg, out)
        st == 0 || error("rnde_node_create: ", unsafe_string(ccall((:rnde_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
        h = new(out[], cfg, 0)
        finalizer(h -> ccall((:rnde_node_destroy, LIB), Cvoid, (Ptr{Cvoid},), h.ptr), h)
        return h
    end
end

check(h::Handle, st) = st == 0 ||
    error("rnde status $st: ", unsafe_string(ccall((:rnde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))

# ---- which of the library's callbacks a caller's `func` is ------------------------------------------------
# The reference hands the layer a closure, `model(x, p1, p2, p3; func = save_func, ...)` (experiments/mnist_node.jl:134, mnist_nsde.jl), and the
# SavingCallback calls it on the integrator after every accepted step (src/models/neural_ode.jl:126-127).  The library computes the value inside
# its kernels instead, so the closure has to be RECOGNISED, not called per step: it is evaluated on two mock integrators and the two values are
# matched against the reference's own callbacks (mnist_node.jl:67 EEst*dt; :74-79 |eigen_est|/stability_size; :88-97 their blend with 0.1;
# neural_ode.jl:54 the constant 0).  Anything else is refused -- a wrong regulariser must not be trained on silently.
struct MockIntegrator
    EEst::Float32
    dt::Float32
    eigen_est::Float32
end
const REG_NONE, REG_ERR, REG_STIFF, REG_ERR_STIFF, REG_STIFF_DT = 0, 1, 2, 3, 4
const _PROBES = (MockIntegrator(2f0, 3f0, 5f0), MockIntegrator(0.5f0, 0.25f0, -7f0))

"""
    reg_code(func, stability_size) -> REG_NONE | REG_ERR | REG_STIFF | REG_ERR_STIFF | REG_STIFF_DT

`stability_size`: `alg_stability_size` of the solver the experiment divides by (Tsit5: 3.5068, SOSRI2: 10.6).
"""
function reg_code(func, stability_size::Real)
    s = Float64(stability_size)
    got = map(m -> Float64(data(func(nothing, 0f0, m))), _PROBES)
    want = Dict(REG_NONE => m -> 0.0,
                REG_ERR => m -> Float64(m.EEst) * m.dt,
                REG_STIFF => m -> abs(Float64(m.eigen_est)) / s,
                REG_ERR_STIFF => m -> Float64(m.EEst) * m.dt + 0.1 * Float64(m.eigen_est) / s,
                REG_STIFF_DT => m -> abs(Float64(m.eigen_est) * m.dt))      # the reference's own test: abs(integrator.eigen_est * integrator.dt), test/test_node.jl:75,:84
    for code in (REG_NONE, REG_ERR, REG_STIFF, REG_ERR_STIFF, REG_STIFF_DT)
        all(isapprox(g, want[code](m); rtol = 1e-4, atol = 1e-7) for (g, m) in zip(got, _PROBES)) && return code
    end
    error("RNDE: `func` is none of the callbacks librnde.so computes (EEst*dt, |eigen_est|/stability_size, EEst*dt + 0.1*eigen_est/stability_size, |eigen_est*dt|, 0): ",
          "on (EEst, dt, eigen_est) = (2, 3, 5) and (0.5, 0.25, -7) it returned ", got)
end

# the solver object the layer was built with (`n.args[1]`): its name, and whether it is the composite whose steps fill `integrator.eigen_est`
# (AutoTsit5(Tsit5()) / AutoSOSRI2(SOSRI2()): experiments/mnist_node.jl:81,:99, mnist_nsde.jl:60).  Plain algorithms leave eigen_est at 0.
function solver_name(args)
    isempty(args) && return :Tsit5, false
    alg = args[1]
    hasproperty(alg, :algs) && return nameof(typeof(alg.algs[1])), true
    return nameof(typeof(alg)), false
end

# the callback code a (func, solver) pair runs with: `integrator.eigen_est` is filled by the composite solvers only; under a plain one it keeps its initial
# value (1 [RECALL], not 0), so the reference would record a constant there -- every callback that reads it is refused, the blend included
function effective_reg(code::Int, composite::Bool)
    composite && return code
    (code == REG_STIFF || code == REG_STIFF_DT || code == REG_ERR_STIFF) && error("RNDE: `func` reads integrator.eigen_est, which only the composite solvers (AutoTsit5 / AutoSOSRI2) fill; ",
                               "with a plain solver the reference records its initial value, a constant -- build the layer with the composite solver")
    return code
end

# Flux.Dense chain (MLPDynamics / TDChain) -> config.  `p` from Flux.destructure is accepted as is.
function config_for(dims::Vector{Int}, acts::Vector{Int}; time_dep, max_batch, reltol, abstol, regularize,
                    max_attempts = 160, device = 0, pre_act = false)
    d = ntuple(i -> Int32(i <= length(dims) ? dims[i] : 0), 9)
    a = ntuple(i -> Int32(i <= length(acts) ? acts[i] : 0), 8)
    NodeConfig(length(acts), d, a, time_dep, pre_act, max_batch, 0, reltol, abstol, regularize, 1, 1, 1, max_attempts, device, 0, 0, 0, 0)
end

"""
    solve_forward(h, x, p, tspan; keep_tape) -> (u, nfe, saveval)

Replaces `solve(prob, Tsit5(); sensealg, callback, kwargs...)` + `diffeqsol_to_trackedarray` + `sol.destats.nf`
(reference neural_ode.jl:131-142).  x, p: device arrays (Float32, column-major D x B / flat).
"""
function solve_forward(h::Handle, x::ROCMatrix{Float32}, p::ROCVector{Float32}, tspan; keep_tape::Bool)
    D, B = size(x)
    u = similar(x)
    nfe = Ref{Int64}(0)
    nsv = Ref{Int32}(0)
    sv = Vector{Float32}(undef, h.cfg.max_attempts + 1)
    GC.@preserve x p u sv begin
        st = ccall((:rnde_node_forward, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float32, Float32, Ptr{Cvoid}, Ref{Int64},
                    Ptr{Float32}, Ref{Int32}, Int32, Ptr{Cvoid}),
                   h.ptr, devptr(x), devptr(p), B, Float32(tspan[1]), Float32(tspan[2]), devptr(u), nfe,
                   sv, nsv, keep_tape ? 1 : 0, _stream())
        check(h, st)
    end
    return u, Int(nfe[]), sv[1:nsv[]]
end

"""
    solve_forward_saveat(h, x, p, tspan, saveat; keep_tape) -> (u3, nfe, saveval)

The {R,true} call methods (reference neural_ode.jl:79-108, :146-180): `u3` is the D x T x B array
`diffeqsol_to_3dtrackedarray` builds (src/utils.jl:17-19), T = length(saveat).
"""
function solve_forward_saveat(h::Handle, x::ROCMatrix{Float32}, p::ROCVector{Float32}, tspan, saveat::Vector{Float32};
                              keep_tape::Bool)
    D, B = size(x)
    u3 = ROCArray{Float32}(undef, D, length(saveat), B)
    nfe = Ref{Int64}(0)
    nsv = Ref{Int32}(0)
    sv = Vector{Float32}(undef, h.cfg.max_attempts + 1)
    GC.@preserve x p u3 sv saveat begin
        st = ccall((:rnde_node_forward_saveat, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float32, Float32, Ptr{Float32}, Int32, Ptr{Cvoid},
                    Ref{Int64}, Ptr{Float32}, Ref{Int32}, Int32, Ptr{Cvoid}),
                   h.ptr, devptr(x), devptr(p), B, Float32(tspan[1]), Float32(tspan[2]), saveat, length(saveat),
                   devptr(u3), nfe, sv, nsv, keep_tape ? 1 : 0, _stream())
        check(h, st)
    end
    return u3, Int(nfe[]), sv[1:nsv[]]
end

# save_everystep = true (neural_ode.jl:10-11): the state after every accepted step (the initial one first when save_start), D x n x B with n
# known after the call (the library solves twice: the step sequence, then the same solve saving at those step ends); backward as after saveat
function solve_forward_everystep(h::Handle, x::ROCMatrix{Float32}, p::ROCVector{Float32}, tspan; save_start::Bool = true, keep_tape::Bool)
    D, B = size(x)
    cap = h.cfg.max_attempts + 1
    buf = ROCArray{Float32}(undef, D * cap * B)
    ts = Vector{Float32}(undef, cap)
    n = Ref{Int32}(0)
    nfe = Ref{Int64}(0)
    nsv = Ref{Int32}(0)
    sv = Vector{Float32}(undef, h.cfg.max_attempts + 1)
    GC.@preserve x p buf sv ts begin
        st = ccall((:rnde_node_forward_everystep, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float32, Float32, Int32, Ptr{Cvoid}, Int32, Ptr{Float32}, Ref{Int32},
                    Ref{Int64}, Ptr{Float32}, Ref{Int32}, Int32, Ptr{Cvoid}),
                   h.ptr, devptr(x), devptr(p), B, Float32(tspan[1]), Float32(tspan[2]), save_start ? 1 : 0, devptr(buf), cap, ts, n,
                   nfe, sv, nsv, keep_tape ? 1 : 0, _stream())
        check(h, st)
    end
    u3 = reshape(buf[1:D * Int(n[]) * B], D, Int(n[]), B)
    return u3, ts[1:n[]], Int(nfe[]), sv[1:nsv[]]
end

# ubar: D x B after solve_forward, D x T x B after solve_forward_saveat; x-bar is D x B either way
function solve_backward(h::Handle, ubar::ROCArray{Float32}, svbar::Vector{Float32}, np::Int)
    xbar = ROCArray{Float32}(undef, size(ubar, 1), size(ubar, ndims(ubar)))
    pbar = ROCArray{Float32}(undef, np)
    tsbar = zeros(Float32, 2)
    GC.@preserve ubar xbar pbar svbar tsbar begin
        st = ccall((:rnde_node_backward, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}),
                   h.ptr, devptr(ubar), svbar, devptr(xbar), devptr(pbar), tsbar, _stream())
        check(h, st)
    end
    return xbar, pbar, tsbar
end

# ---- Tracker glue: one tape node for the whole solve ------------------------------------------------
# rnde_solve(h, x, p, tspan) returns (u, saveval); nfe is kept on the handle (non-differentiable).
rnde_solve(h::Handle, x::TrackedArray, p::TrackedArray, tspan) = track(rnde_solve, h, x, p, tspan)

@grad function rnde_solve(h::Handle, x, p, tspan)
    u, nfe, sv = solve_forward(h, data(x), data(p), data.(tspan); keep_tape = true)
    h.last_nfe = nfe
    return (u, sv), function (Δ)
        ubar, svbar = Δ
        xbar, pbar, tsbar = solve_backward(h, ROCArray{Float32}(ubar), Vector{Float32}(svbar), length(p))
        return (nothing, xbar, pbar, tsbar)
    end
end

# the {R,true} methods (reference neural_ode.jl:79-108, :146-180): u3 is D x T x B, its cotangent has the same shape
rnde_solve_saveat(h::Handle, x::TrackedArray, p::TrackedArray, tspan, saveat::Vector{Float32}) = track(rnde_solve_saveat, h, x, p, tspan, saveat)

@grad function rnde_solve_saveat(h::Handle, x, p, tspan, saveat)
    u3, nfe, sv = solve_forward_saveat(h, data(x), data(p), data.(tspan), saveat; keep_tape = true)
    h.last_nfe = nfe
    return (u3, sv), function (Δ)
        ubar, svbar = Δ
        xbar, pbar, tsbar = solve_backward(h, ROCArray{Float32}(ubar), Vector{Float32}(svbar), length(p))
        return (nothing, xbar, pbar, tsbar, nothing)
    end
end

# The unregularised methods ({false,false} / {false,true}) go through the same two rules: their handle is created with regularize = 0, the
# saved-value vector then comes back empty and its cotangent is ignored.

# ---- what changes in src/models/neural_ode.jl: bindings/julia/patch_neural_ode.jl holds the four call methods as real method
# definitions (include it after `using RegNeuralDE`); the body between ODEProblem construction and result unpacking (reference
# :126-138) is the one call below, everything else -- signature, keyword defaults, `_convert_tspan`, the returned triple -- is the reference's:
#
#
#   @fastmath function (n::TrackedNeuralODE{true,false})(x, p = n.p; func = ..., tspan = nothing, saveat = nothing)
#       tspan = _convert_tspan(isnothing(tspan) ? n.tspan : tspan, p)
#       res, saveval = RNDE.rnde_solve(n.rnde_handle, x, p, tspan)        # <- replaces :126-138
#       sv = SavedValues(eltype(tspan), eltype(p)); append!(sv.saveval, saveval)
#       return res, n.rnde_handle.last_nfe, sv
#   end
#
# and the constructor (:10-33) creates `rnde_handle = RNDE.Handle(RNDE.config_for(...))` from the Dense sizes
# of `model`, kwargs[:reltol], kwargs[:abstol] and `regularize`.

# ---- TrackedNeuralDSDE (reference src/models/neural_sde.jl) -----------------------------------------
struct NsdeConfig   # mirrors rnde_nsde_config
    drift_layers::Int32
    drift_dims::NTuple{9,Int32}
    drift_act::NTuple{8,Int32}
    diff_layers::Int32
    diff_dims::NTuple{9,Int32}
    diff_act::NTuple{8,Int32}
    max_batch::Int32
    solver::Int32          # 0 SOSRI, 1 SRIW1, 2 SOSRI2
    reltol::Float32
    abstol::Float32
    regularize::Int32
    cb_save_start::Int32
    max_attempts::Int32
    device::Int32
    beta1::Float32; beta2::Float32; gamma::Float32; qmin::Float32; qmax::Float32; qoldinit::Float32; delta::Float32
    generic::Int32
    stability_size::Float32      # RNDE_REG_STIFF: 0 = alg_stability_size(SOSRI2()) = 10.6
end

mutable struct NsdeHandle
    ptr::Ptr{Cvoid}
    cfg::NsdeConfig
    function NsdeHandle(cfg::NsdeConfig)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:rnde_nsde_create, LIB), Cint, (Ref{NsdeConfig}, Ref{Ptr{Cvoid}}), cfg, out)
        st == 0 || error("rnde_nsde_create: ", unsafe_string(ccall((:rnde_nsde_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
        h = new(out[], cfg)
        finalizer(h -> ccall((:rnde_nsde_destroy, LIB), Cvoid, (Ptr{Cvoid},), h.ptr), h)
        return h
    end
end

"""
    nsde_forward(h, x, p, tspan; noise = nothing, seed = 0, keep_tape) -> (u, nfe1, nfe2, saveval)

Replaces `solve(prob, SOSRI(); sensealg, callback, kwargs...)` and the unpacking at neural_sde.jl:98-113.
`noise`: a `ROCArray{Float32,4}` of size (D, B, 2, n_pool) filled by `randn!` (the caller's own random stream: memory order
D fastest, then B, then W/Z, then the draw -- exactly the pool layout of include/rnde.h), or `nothing` for the library's
Philox stream named by `seed`.
"""
function nsde_forward(h::NsdeHandle, x::ROCMatrix{Float32}, p::ROCVector{Float32}, tspan; noise = nothing, seed::Integer = 0, keep_tape::Bool)
    D, B = size(x)
    u = similar(x)
    nfe1 = Ref{Int64}(0); nfe2 = Ref{Int64}(0); nsv = Ref{Int32}(0)
    sv = Vector{Float32}(undef, h.cfg.max_attempts + 1)
    npool = noise === nothing ? 0 : size(noise, 4)
    GC.@preserve x p u sv noise begin
        st = ccall((:rnde_nsde_forward, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float32, Float32, Ptr{Cvoid}, Int32, UInt64, Ptr{Cvoid},
                    Ref{Int64}, Ref{Int64}, Ptr{Float32}, Ref{Int32}, Int32, Ptr{Cvoid}),
                   h.ptr, devptr(x), devptr(p), B, Float32(tspan[1]), Float32(tspan[2]),
                   noise === nothing ? C_NULL : devptr(noise), npool, UInt64(seed), devptr(u), nfe1, nfe2, sv, nsv, keep_tape ? 1 : 0, _stream())
        st == 0 || error("rnde_nsde_forward status $st: ", unsafe_string(ccall((:rnde_nsde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))
    end
    return u, Int(nfe1[]), Int(nfe2[]), sv[1:nsv[]]
end

function nsde_backward(h::NsdeHandle, ubar::ROCMatrix{Float32}, svbar::Vector{Float32}, np::Int)
    xbar = similar(ubar)
    pbar = ROCArray{Float32}(undef, np)
    GC.@preserve ubar xbar pbar svbar begin
        st = ccall((:rnde_nsde_backward, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                   h.ptr, devptr(ubar), svbar, devptr(xbar), devptr(pbar), _stream())
        st == 0 || error("rnde_nsde_backward status $st")
    end
    return xbar, pbar
end

function nsde_forward_saveat(h::NsdeHandle, x::ROCMatrix{Float32}, p::ROCVector{Float32}, tspan, saveat::Vector{Float32}; noise = nothing,
                             seed::Integer = 0, keep_tape::Bool)
    D, B = size(x)
    u3 = ROCArray{Float32}(undef, D, length(saveat), B)
    nfe1 = Ref{Int64}(0); nfe2 = Ref{Int64}(0); nsv = Ref{Int32}(0)
    sv = Vector{Float32}(undef, h.cfg.max_attempts + 1)
    npool = noise === nothing ? 0 : size(noise, 4)
    GC.@preserve x p u3 sv noise saveat begin
        st = ccall((:rnde_nsde_forward_saveat, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float32, Float32, Ptr{Cvoid}, Int32, UInt64, Ptr{Float32}, Int32, Ptr{Cvoid},
                    Ref{Int64}, Ref{Int64}, Ptr{Float32}, Ref{Int32}, Int32, Ptr{Cvoid}),
                   h.ptr, devptr(x), devptr(p), B, Float32(tspan[1]), Float32(tspan[2]),
                   noise === nothing ? C_NULL : devptr(noise), npool, UInt64(seed), saveat, length(saveat), devptr(u3), nfe1, nfe2, sv, nsv,
                   keep_tape ? 1 : 0, _stream())
        st == 0 || error("rnde_nsde_forward_saveat status $st: ", unsafe_string(ccall((:rnde_nsde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))
    end
    return u3, Int(nfe1[]), Int(nfe2[]), sv[1:nsv[]]
end

# save_everystep = true of the SDE layer (neural_sde.jl:14): every accepted step's end (t0 first when save_start), two solves on the same noise inside
function nsde_forward_everystep(h::NsdeHandle, x::ROCMatrix{Float32}, p::ROCVector{Float32}, tspan; noise = nothing, seed::Integer = 0,
                                save_start::Bool = true, keep_tape::Bool)
    D, B = size(x)
    cap = h.cfg.max_attempts + 1
    buf = ROCArray{Float32}(undef, D * cap * B)
    ts = Vector{Float32}(undef, cap)
    n = Ref{Int32}(0); nfe1 = Ref{Int64}(0); nfe2 = Ref{Int64}(0); nsv = Ref{Int32}(0)
    sv = Vector{Float32}(undef, h.cfg.max_attempts + 1)
    npool = noise === nothing ? 0 : size(noise, 4)
    GC.@preserve x p buf sv noise ts begin
        st = ccall((:rnde_nsde_forward_everystep, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float32, Float32, Ptr{Cvoid}, Int32, UInt64, Int32, Ptr{Cvoid}, Int32, Ptr{Float32},
                    Ref{Int32}, Ref{Int64}, Ref{Int64}, Ptr{Float32}, Ref{Int32}, Int32, Ptr{Cvoid}),
                   h.ptr, devptr(x), devptr(p), B, Float32(tspan[1]), Float32(tspan[2]),
                   noise === nothing ? C_NULL : devptr(noise), npool, UInt64(seed), save_start ? 1 : 0, devptr(buf), cap, ts, n, nfe1, nfe2, sv, nsv,
                   keep_tape ? 1 : 0, _stream())
        st == 0 || error("rnde_nsde_forward_everystep status $st: ", unsafe_string(ccall((:rnde_nsde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))
    end
    return reshape(buf[1:D * Int(n[]) * B], D, Int(n[]), B), ts[1:n[]], Int(nfe1[]), Int(nfe2[]), sv[1:nsv[]]
end

# u-bar: D x B after nsde_forward, D x T x B after nsde_forward_saveat; x-bar is D x B either way
function nsde_backward_any(h::NsdeHandle, ubar::ROCArray{Float32}, svbar::Vector{Float32}, np::Int)
    xbar = ROCArray{Float32}(undef, size(ubar, 1), size(ubar, ndims(ubar)))
    pbar = ROCArray{Float32}(undef, np)
    GC.@preserve ubar xbar pbar svbar begin
        st = ccall((:rnde_nsde_backward, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                   h.ptr, devptr(ubar), svbar, devptr(xbar), devptr(pbar), _stream())
        st == 0 || error("rnde_nsde_backward status $st: ", unsafe_string(ccall((:rnde_nsde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))
    end
    return xbar, pbar
end

# Tracker glue for the stochastic layer: one tape node for the whole solve.  The evaluation counters the reference keeps in the layer's mutable
# `nfes` vector (neural_sde.jl:11, :46, :50, :142-143) are kept on the handle.  `seed` names the library's Philox stream (a caller that wants
# Julia's own random stream passes a pool of normals through nsde_forward directly).
mutable struct NsdeCounters; nfe1::Int; nfe2::Int; end
const NSDE_COUNTERS = IdDict{NsdeHandle,NsdeCounters}()
counters(h::NsdeHandle) = get!(() -> NsdeCounters(0, 0), NSDE_COUNTERS, h)

rnde_nsde_solve(h::NsdeHandle, x::TrackedArray, p::TrackedArray, tspan, seed::Integer) = track(rnde_nsde_solve, h, x, p, tspan, seed)
@grad function rnde_nsde_solve(h::NsdeHandle, x, p, tspan, seed)
    u, nfe1, nfe2, sv = nsde_forward(h, data(x), data(p), data.(tspan); seed = seed, keep_tape = true)
    c = counters(h); c.nfe1 = nfe1; c.nfe2 = nfe2
    return (u, sv), function (Δ)
        ubar, svbar = Δ
        xbar, pbar = nsde_backward_any(h, ROCArray{Float32}(ubar), Vector{Float32}(svbar), length(p))
        return (nothing, xbar, pbar, nothing, nothing)      # (the SDE step-size controller strips tracking: no tspan cotangent, DESIGN.md 4.3)
    end
end

rnde_nsde_solve_saveat(h::NsdeHandle, x::TrackedArray, p::TrackedArray, tspan, saveat::Vector{Float32}, seed::Integer) =
    track(rnde_nsde_solve_saveat, h, x, p, tspan, saveat, seed)
@grad function rnde_nsde_solve_saveat(h::NsdeHandle, x, p, tspan, saveat, seed)
    u3, nfe1, nfe2, sv = nsde_forward_saveat(h, data(x), data(p), data.(tspan), saveat; seed = seed, keep_tape = true)
    c = counters(h); c.nfe1 = nfe1; c.nfe2 = nfe2
    return (u3, sv), function (Δ)
        ubar, svbar = Δ
        xbar, pbar = nsde_backward_any(h, ROCArray{Float32}(ubar), Vector{Float32}(svbar), length(p))
        return (nothing, xbar, pbar, nothing, nothing, nothing)
    end
end

# config from the two Flux chains of TrackedNeuralDSDE (neural_sde.jl:13-41): Dense sizes and activations, tolerances from kwargs
function nsde_config_for(drift_dims::Vector{Int}, drift_acts::Vector{Int}, diff_dims::Vector{Int}, diff_acts::Vector{Int}; max_batch, reltol, abstol,
                         regularize, solver = 0, max_attempts = 256, device = 0)
    t9(v) = ntuple(i -> Int32(i <= length(v) ? v[i] : 0), 9)
    t8(v) = ntuple(i -> Int32(i <= length(v) ? v[i] : 0), 8)
    NsdeConfig(length(drift_acts), t9(drift_dims), t8(drift_acts), length(diff_acts), t9(diff_dims), t8(diff_acts), max_batch, solver, reltol, abstol,
               regularize, 1, max_attempts, device, 0f0, 0f0, 0f0, 0f0, 0f0, 0f0, 0f0, 0, 0f0)
end

# ---- one training-step gradient in one call ------------------------------------------------------------
# loss_function + Tracker.gradient of experiments/mnist_node.jl:132-137, :229-233 for ClassifierNODE with preode = identity:
# forward solve (taped) -> Dense(D, C) + logitcrossentropy and their reverse -> reverse solve, with the head queued before the
# forward's host wait.  Returns (ce (1-element device array), lambda * mean(saveval), nfe); the gradients land in p2bar / p3bar
# in stream order.  comm: the handle of comm_create (C_NULL: single GPU) -- both gradients are then sum-all-reduced in place.
function classifier_grad!(p2bar::ROCVector{Float32}, p3bar::ROCVector{Float32}, h::Handle, x::ROCMatrix{Float32},
                          p2::ROCVector{Float32}, p3::ROCVector{Float32}, y::ROCMatrix{Float32}, tspan; lambda = 1f2,
                          comm::Ptr{Cvoid} = C_NULL)
    ce = ROCVector{Float32}(undef, 1)
    reg = Ref{Cfloat}(0); nfe = Ref{Int64}(0)
    st = GC.@preserve p2bar p3bar x p2 p3 y ce begin
        ccall((:rnde_node_classifier_grad, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Cfloat, Cfloat, Cfloat,
               Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cfloat}, Ref{Int64}, Ptr{Cvoid}, Ptr{Cvoid}),
              h.ptr, devptr(x), devptr(p2), devptr(p3), devptr(y), size(x, 2), size(y, 1), Float32(tspan[1]), Float32(tspan[2]),
              Float32(lambda), devptr(p2bar), devptr(p3bar), C_NULL, devptr(ce), reg, nfe, comm, _stream())
    end
    st == 0 || error("rnde_node_classifier_grad: ", unsafe_string(ccall((:rnde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))
    return ce, reg[], nfe[]
end

# The SDE counterpart (ClassifierNSDE around the solve, one trajectory per input): x is the SDE's initial state (the presde layer's output),
# xbar its cotangent; noise = nothing: the library's Philox stream named by `seed`, else the caller's pool (D x B x 2 x n_pool normals).
function nsde_classifier_grad!(p2bar::ROCVector{Float32}, p3bar::ROCVector{Float32}, xbar::ROCMatrix{Float32}, h::NsdeHandle,
                               x::ROCMatrix{Float32}, p2::ROCVector{Float32}, p3::ROCVector{Float32}, y::ROCMatrix{Float32}, tspan;
                               lambda = 1f1, noise = nothing, seed::Integer = 0)
    ce = ROCVector{Float32}(undef, 1)
    reg = Ref{Cfloat}(0); nfe1 = Ref{Int64}(0); nfe2 = Ref{Int64}(0)
    npool = noise === nothing ? 0 : size(noise, 4)
    st = GC.@preserve p2bar p3bar xbar x p2 p3 y ce noise begin
        ccall((:rnde_nsde_classifier_grad, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Cfloat, Cfloat, Ptr{Cvoid}, Int32, UInt64, Cfloat,
               Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cfloat}, Ref{Int64}, Ref{Int64}, Ptr{Cvoid}),
              h.ptr, devptr(x), devptr(p2), devptr(p3), devptr(y), size(x, 2), size(y, 1), Float32(tspan[1]), Float32(tspan[2]),
              noise === nothing ? C_NULL : devptr(noise), npool, UInt64(seed), Float32(lambda),
              devptr(p2bar), devptr(p3bar), devptr(xbar), devptr(ce), reg, nfe1, nfe2, _stream())
    end
    st == 0 || error("rnde_nsde_classifier_grad: ", unsafe_string(ccall((:rnde_nsde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))
    return ce, reg[], nfe1[], nfe2[]
end

# ---- optimiser step and the data-parallel collective -------------------------------------------------
# Optimiser(InvDecay(gamma), Momentum(eta, rho)) on one flat group, in place (src/utils.jl:149-156, mnist_node.jl:130);
# `n` is the group's InvDecay counter (starts at 1, the caller increments it); gscale = 1 / nworkers after a summed all-reduce.
function momentum_step!(p::ROCVector{Float32}, g::ROCVector{Float32}, v::ROCVector{Float32}, n::Integer;
                        gamma = 1f-5, eta = 0.1f0, rho = 0.9f0, gscale = 1f0)
    GC.@preserve p g v begin
        st = ccall((:rnde_momentum_step_scaled, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cfloat, Cfloat, Cfloat, Cfloat, Ptr{Cvoid}),
                   devptr(p), devptr(g), devptr(v), length(p), n, gamma, eta, rho, gscale, _stream())
    end
    st == 0 || error("rnde_momentum_step_scaled: status $st")
    return p
end

# Flux.Optimise.ADAM(eta, (beta1, beta2)) on one flat parameter group (experiments/mnist_nsde.jl): m, v are the caller's state arrays (zeros at t = 1)
function adam_step!(p::ROCVector{Float32}, g::ROCVector{Float32}, m::ROCVector{Float32}, v::ROCVector{Float32}, t::Integer;
                    eta = 0.001f0, beta = (0.9f0, 0.999f0), eps = 1f-8, gscale = 1f0)
    st = GC.@preserve p g m v begin
        ccall((:rnde_adam_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cfloat, Cfloat, Cfloat, Cfloat, Cfloat, Ptr{Cvoid}),
              devptr(p), devptr(g), devptr(m), devptr(v), length(p), t, eta, beta[1], beta[2], eps, gscale, _stream())
    end
    st == 0 || error("rnde_adam_step: status $st")
    return p
end

# one process per GPU (Distributed / MPI.jl carry the 128-byte id from rank 0 to the others)
comm_unique_id() = (id = zeros(UInt8, 128); ccall((:rnde_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id) == 0 || error("rnde_comm_unique_id"); id)
function comm_create(id::Vector{UInt8}, rank::Integer, world::Integer, device::Integer)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    ccall((:rnde_comm_create, LIB), Cint, (Ptr{UInt8}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}), id, rank, world, device, out) == 0 ||
        error("rnde_comm_create: ", unsafe_string(ccall((:rnde_comm_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    return out[]
end
allreduce_sum!(comm::Ptr{Cvoid}, g::ROCVector{Float32}) = GC.@preserve g begin
    ccall((:rnde_comm_allreduce, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}), comm, devptr(g), length(g), 0, _stream()) == 0 ||
        error("rnde_comm_allreduce")
    g
end

# The same all-reduce as ONE kernel over peer-mapped windows (opt-in, include/rnde.h): every rank makes a window, the 64-byte handles travel to
# all ranks in rank order (Distributed / MPI.jl: an all-gather of 64 bytes), comm_create_peers maps them.  `ENV["RNDE_ONESHOT"] = "1"` in front of
# comm_create does the same through RCCL's own all-gather.  comm_path says which path a communicator's all-reduces take.
function comm_window_create(device::Integer)
    win = Ref{Ptr{Cvoid}}(C_NULL); handle = zeros(UInt8, 64)
    ccall((:rnde_comm_window_create, LIB), Cint, (Int32, Ref{Ptr{Cvoid}}, Ptr{UInt8}), device, win, handle) == 0 ||
        error("rnde_comm_window_create: ", unsafe_string(ccall((:rnde_comm_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    return win[], handle
end
function comm_create_peers(win::Ptr{Cvoid}, handles::Vector{UInt8}, rank::Integer, world::Integer)
    length(handles) == 64 * world || error("comm_create_peers: 64 bytes per rank, in rank order")
    out = Ref{Ptr{Cvoid}}(C_NULL)
    ccall((:rnde_comm_create_peers, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32, Ref{Ptr{Cvoid}}), win, handles, rank, world, out) == 0 ||
        error("rnde_comm_create_peers: ", unsafe_string(ccall((:rnde_comm_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    return out[]
end
comm_path(comm::Ptr{Cvoid}) = unsafe_string(ccall((:rnde_comm_path, LIB), Cstring, (Ptr{Cvoid},), comm))

# SURVEY 8e mode 2: ONE step-size controller for all column shards of a minibatch (rnde_node_set_coupling): every rank calls it with
# its communicator and the global batch; `comm = C_NULL` returns to independent controllers.  Equal shards, the same calls on every rank.
function set_coupling!(h::Handle, comm::Ptr{Cvoid}, global_batch::Integer)
    st = ccall((:rnde_node_set_coupling, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32), h.ptr, comm, global_batch)
    st == 0 || error("rnde_node_set_coupling: ", unsafe_string(ccall((:rnde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))
    return h
end

# Which unit forms the Dense-layer products of the one-launch forward solve (include/rnde.h: rnde_node_set_matrix_mode):
# 0 = fp32-input MFMA (vector ALUs on gfx950), 1 = exact three-way bf16 split on the matrix cores (the default where the kernels serve the shape).
function set_matrix_mode!(h::Handle, mode::Integer)
    st = ccall((:rnde_node_set_matrix_mode, LIB), Cint, (Ptr{Cvoid}, Cint), h.ptr, Cint(mode))
    check(h, st)
end
matrix_mode(h::Handle) = Int(ccall((:rnde_node_matrix_mode, LIB), Cint, (Ptr{Cvoid},), h.ptr))

end # module
