# RNDE.jl -- Julia binding of librnde.so (include/rnde.h) for RegNeuralDE.jl.
#
# SOURCE ONLY: no Julia toolchain exists in the build image, so this file has never been executed.
# It shows exactly what a maintainer adds to the reference: the body of the TrackedNeuralODE call methods
# between ODEProblem construction and result unpacking (reference src/models/neural_ode.jl:126-142) becomes
# one `ccall`, and the reverse sweep is registered with `Tracker.@grad` so that
# `Tracker.gradient(...)` (reference experiments/mnist_node.jl:229-232) keeps working unchanged.
module RNDE

using CUDA: CuArray           # the reference's array type; on MI355X use AMDGPU.ROCArray (same pointer semantics)
using Tracker
using Tracker: TrackedArray, data, track, @grad

const LIB = joinpath(@__DIR__, "..", "..", "regneuralde.jl_amd", "lib", "librnde.so")
const MAX_LAYERS = 8

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
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    cfg::NodeConfig
    function Handle(cfg::NodeConfig)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:rnde_node_create, LIB), Cint, (Ref{NodeConfig}, Ref{Ptr{Cvoid}}), cfg, out)
        st == 0 || error("rnde_node_create: ", unsafe_string(ccall((:rnde_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
        h = new(out[], cfg)
        finalizer(h -> ccall((:rnde_node_destroy, LIB), Cvoid, (Ptr{Cvoid},), h.ptr), h)
        return h
    end
end

check(h::Handle, st) = st == 0 ||
    error("rnde status $st: ", unsafe_string(ccall((:rnde_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)))

# Flux.Dense chain (MLPDynamics / TDChain) -> config.  `p` from Flux.destructure is accepted as is.
function config_for(dims::Vector{Int}, acts::Vector{Int}; time_dep, max_batch, reltol, abstol, regularize,
                    max_attempts = 160, device = 0)
    d = ntuple(i -> Int32(i <= length(dims) ? dims[i] : 0), 9)
    a = ntuple(i -> Int32(i <= length(acts) ? acts[i] : 0), 8)
    NodeConfig(length(acts), d, a, time_dep, 0, max_batch, 0, reltol, abstol, regularize, 1, 1, 1, max_attempts, device, 0)
end

"""
    solve_forward(h, x, p, tspan; keep_tape) -> (u, nfe, saveval)

Replaces `solve(prob, Tsit5(); sensealg, callback, kwargs...)` + `diffeqsol_to_trackedarray` + `sol.destats.nf`
(reference neural_ode.jl:131-142).  x, p: device arrays (Float32, column-major D x B / flat).
"""
function solve_forward(h::Handle, x::CuArray{Float32,2}, p::CuArray{Float32,1}, tspan; keep_tape::Bool)
    D, B = size(x)
    u = similar(x)
    nfe = Ref{Int64}(0)
    nsv = Ref{Int32}(0)
    sv = Vector{Float32}(undef, h.cfg.max_attempts + 1)
    GC.@preserve x p u sv begin
        st = ccall((:rnde_node_forward, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Ptr{Float32}, Ref{Int64},
                    Ptr{Float32}, Ref{Int32}, Int32, Ptr{Cvoid}),
                   h.ptr, pointer(x), pointer(p), B, Float32(tspan[1]), Float32(tspan[2]), pointer(u), nfe,
                   sv, nsv, keep_tape ? 1 : 0, C_NULL)
        check(h, st)
    end
    return u, Int(nfe[]), sv[1:nsv[]]
end

"""
    solve_forward_saveat(h, x, p, tspan, saveat; keep_tape) -> (u3, nfe, saveval)

The {R,true} call methods (reference neural_ode.jl:79-108, :146-180): `u3` is the D x T x B array
`diffeqsol_to_3dtrackedarray` builds (src/utils.jl:17-19), T = length(saveat).
"""
function solve_forward_saveat(h::Handle, x::CuArray{Float32,2}, p::CuArray{Float32,1}, tspan, saveat::Vector{Float32};
                              keep_tape::Bool)
    D, B = size(x)
    u3 = CuArray{Float32}(undef, D, length(saveat), B)
    nfe = Ref{Int64}(0)
    nsv = Ref{Int32}(0)
    sv = Vector{Float32}(undef, h.cfg.max_attempts + 1)
    GC.@preserve x p u3 sv saveat begin
        st = ccall((:rnde_node_forward_saveat, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Ptr{Float32}, Int32, Ptr{Float32},
                    Ref{Int64}, Ptr{Float32}, Ref{Int32}, Int32, Ptr{Cvoid}),
                   h.ptr, pointer(x), pointer(p), B, Float32(tspan[1]), Float32(tspan[2]), saveat, length(saveat),
                   pointer(u3), nfe, sv, nsv, keep_tape ? 1 : 0, C_NULL)
        check(h, st)
    end
    return u3, Int(nfe[]), sv[1:nsv[]]
end

# ubar: D x B after solve_forward, D x T x B after solve_forward_saveat; x-bar is D x B either way
function solve_backward(h::Handle, ubar::CuArray{Float32}, svbar::Vector{Float32}, np::Int)
    xbar = CuArray{Float32}(undef, size(ubar, 1), size(ubar, ndims(ubar)))
    pbar = CuArray{Float32}(undef, np)
    tsbar = zeros(Float32, 2)
    GC.@preserve ubar xbar pbar svbar tsbar begin
        st = ccall((:rnde_node_backward, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
                   h.ptr, pointer(ubar), svbar, pointer(xbar), pointer(pbar), tsbar, C_NULL)
        check(h, st)
    end
    return xbar, pbar, tsbar
end

# ---- Tracker glue: one tape node for the whole solve ------------------------------------------------
# rnde_solve(h, x, p, tspan) returns (u, saveval); nfe is stashed on the handle side (non-differentiable).
rnde_solve(h::Handle, x::TrackedArray, p::TrackedArray, tspan) = track(rnde_solve, h, x, p, tspan)

@grad function rnde_solve(h::Handle, x, p, tspan)
    u, nfe, sv = solve_forward(h, data(x), data(p), data.(tspan); keep_tape = true)
    LAST_NFE[] = nfe
    return (u, sv), function (Δ)
        ubar, svbar = Δ
        xbar, pbar, tsbar = solve_backward(h, CuArray{Float32,2}(ubar), Vector{Float32}(svbar), length(p))
        return (nothing, xbar, pbar, tsbar)
    end
end
const LAST_NFE = Ref(0)

# ---- what changes in src/models/neural_ode.jl (reference :110-144) -----------------------------------
#
#   @fastmath function (n::TrackedNeuralODE{true,false})(x, p = n.p; func = ..., tspan = nothing, saveat = nothing)
#       tspan = _convert_tspan(isnothing(tspan) ? n.tspan : tspan, p)
#       res, saveval = RNDE.rnde_solve(n.rnde_handle, x, p, tspan)        # <- replaces :126-138
#       sv = SavedValues(eltype(tspan), eltype(p)); append!(sv.saveval, saveval)
#       return res, RNDE.LAST_NFE[], sv
#   end
#
# and the constructor (:10-33) creates `rnde_handle = RNDE.Handle(RNDE.config_for(...))` from the Dense sizes
# of `model`, kwargs[:reltol], kwargs[:abstol] and `regularize`.

# Optimiser(InvDecay(gamma), Momentum(eta, rho)) on one flat group, in place (src/utils.jl:149-156, mnist_node.jl:130);
# `n` is the group's InvDecay counter (starts at 1, the caller increments it).
function momentum_step!(p::CuArray{Float32,1}, g::CuArray{Float32,1}, v::CuArray{Float32,1}, n::Integer;
                        gamma = 1f-5, eta = 0.1f0, rho = 0.9f0)
    GC.@preserve p g v begin
        st = ccall((:rnde_momentum_step, LIB), Cint,
                   (CuPtr{Cfloat}, CuPtr{Cfloat}, CuPtr{Cfloat}, Int64, Int64, Cfloat, Cfloat, Cfloat, Ptr{Cvoid}),
                   p, g, v, length(p), n, gamma, eta, rho, C_NULL)
    end
    st == 0 || error("rnde_momentum_step: status $st")
    return p
end

end # module
