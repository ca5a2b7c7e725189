"""regneuralde.jl_amd -- MI355X-native drop-in for RegNeuralDE.jl's adaptive Runge-Kutta hot path.

Only the hot path lives here: csrc/ (HIP kernels + C ABI, built into lib/librnde.so) and the
host-side mirror of the reference's layer interface.  See DESIGN.md and INTEGRATION.md.
"""
from . import _lib, build  # noqa: F401
from .layers import Chain, Dense, LatentGenDynamics, MLPDynamics, TDChain, destructure  # noqa: F401
from .node import SavedValues, TrackedNeuralODE  # noqa: F401
from .nsde import ClassifierNSDE, TrackedNeuralDSDE, fused_nsde_loss_and_grad, nsde_loss_function  # noqa: F401
from .classifier import ClassifierNODE, FluxADAM, FluxOptimiser, accuracy, fused_loss_and_grad, REGULARISERS, lambda_schedule, logitcrossentropy, loss_function, sample_tspan_ubound  # noqa: F401
from .dataparallel import FlatGrads, GradientAllReducer, shard_columns  # noqa: F401
from .timeseries import FluxAdaMax, fused_latent_loss_and_grad, LatentGRU, LatentTimeSeriesModel, build_latent_ode, get_t_saveat, kl_divergence, lambda_k, latent_loss_function, log_likelihood, sample_tbounds  # noqa: F401
