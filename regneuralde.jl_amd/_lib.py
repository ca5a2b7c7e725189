"""ctypes binding of include/rnde.h.  Fails loudly when librnde.so is missing: there is no CPU path."""
import ctypes as C
import os

from . import build as _build

MAX_LAYERS = 8
OK, BAD_ARG, MAX_ATTEMPTS, DT_UNDERFLOW, NONFINITE, HIP_ERR, NO_TAPE, NO_DEVICE = range(8)
REG = {"none": 0, False: 0, None: 0, "error_est": 1, True: 1, "stiff_est": 2, "error_stiff_est": 3}


class NodeConfig(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dims", C.c_int32 * (MAX_LAYERS + 1)), ("act", C.c_int32 * MAX_LAYERS),
                ("time_dep", C.c_int32), ("pre_act", C.c_int32), ("max_batch", C.c_int32), ("solver", C.c_int32),
                ("reltol", C.c_float), ("abstol", C.c_float), ("regularize", C.c_int32),
                ("cb_save_start", C.c_int32), ("track_ctrl", C.c_int32), ("track_initdt", C.c_int32),
                ("max_attempts", C.c_int32), ("device", C.c_int32), ("col_tile", C.c_int32),
                ("persist", C.c_int32), ("wgrad_side_pct", C.c_int32), ("stage_generic", C.c_int32)]


class NsdeConfig(C.Structure):
    _fields_ = [("drift_layers", C.c_int32), ("drift_dims", C.c_int32 * (MAX_LAYERS + 1)), ("drift_act", C.c_int32 * MAX_LAYERS),
                ("diff_layers", C.c_int32), ("diff_dims", C.c_int32 * (MAX_LAYERS + 1)), ("diff_act", C.c_int32 * MAX_LAYERS),
                ("max_batch", C.c_int32), ("solver", C.c_int32), ("reltol", C.c_float), ("abstol", C.c_float),
                ("regularize", C.c_int32), ("cb_save_start", C.c_int32), ("max_attempts", C.c_int32), ("device", C.c_int32),
                ("beta1", C.c_float), ("beta2", C.c_float), ("gamma", C.c_float), ("qmin", C.c_float), ("qmax", C.c_float),
                ("qoldinit", C.c_float), ("delta", C.c_float), ("generic", C.c_int32), ("stability_size", C.c_float)]


class LatentConfig(C.Structure):
    _fields_ = [("max_batch", C.c_int32), ("max_T", C.c_int32), ("device", C.c_int32)]


ODE_SOLVER = {"Tsit5": 0, "AutoTsit5": 0, "DP5": 1, "DOP853": 2}
SDE_SOLVER = {"SOSRI": 0, "SRIW1": 1, "SOSRI2": 2, "AutoSOSRI2": 2}      # AutoSOSRI2(SOSRI2()) (mnist_nsde.jl:60): SOSRI2's trajectory + integrator.eigen_est


class RndeError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"rnde status {status}: {msg}")
        self.status = status


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback for the integration path.")
    L = C.CDLL(path)
    vp, f, i32, i64p, fp, i32p = C.c_void_p, C.c_float, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.rnde_version.restype = C.c_char_p
    L.rnde_last_error.restype = C.c_char_p
    L.rnde_last_error.argtypes = [vp]
    L.rnde_param_count.restype = i32
    L.rnde_param_count.argtypes = [C.POINTER(NodeConfig)]
    L.rnde_node_create.argtypes = [C.POINTER(NodeConfig), C.POINTER(vp)]
    L.rnde_node_destroy.argtypes = [vp]
    L.rnde_node_destroy.restype = None
    L.rnde_node_forward.argtypes = [vp, vp, vp, i32, f, f, vp, i64p, fp, i32p, i32, vp]
    L.rnde_node_forward_saveat.argtypes = [vp, vp, vp, i32, f, f, fp, i32, vp, i64p, fp, i32p, i32, vp]
    L.rnde_node_forward_everystep.argtypes = [vp, vp, vp, i32, f, f, i32, vp, i32, fp, i32p, i64p, fp, i32p, i32, vp]
    L.rnde_node_forward_replay.argtypes = [vp, vp, vp, i32, f, f, fp, i32, vp, i64p, fp, i32p, i32, vp]
    L.rnde_node_backward.argtypes = [vp, vp, fp, vp, vp, fp, vp]
    L.rnde_node_backward_async.restype = C.c_int32
    L.rnde_node_backward_async.argtypes = [vp, vp, fp, vp, vp, vp, vp]
    L.rnde_node_release_tape.argtypes = [vp]
    L.rnde_node_forward_host.argtypes = [vp, fp, fp, i32, f, f, fp, i64p, fp, i32p, i32]
    L.rnde_node_backward_host.argtypes = [vp, fp, fp, fp, fp, fp]
    L.rnde_node_steps.argtypes = [vp, fp, i32, i32p]
    L.rnde_debug_feval.argtypes = [vp, vp, vp, i32, f, vp, vp]
    L.rnde_debug_attempt.argtypes = [vp, vp, vp, vp, i32, f, f, vp, vp, fp, vp]
    L.rnde_bench_attempt.argtypes = [vp, vp, vp, i32, i32, fp, vp]
    L.rnde_bench_attempt_taped.argtypes = [vp, vp, vp, i32, i32, fp, vp]
    L.rnde_bench_attempt_cold_tape.argtypes = [vp, vp, vp, i32, i32, i32, fp, vp]
    L.rnde_node_set_timing.argtypes = [vp, i32]
    L.rnde_node_timing.argtypes = [vp, fp, fp, fp]
    L.rnde_node_fallback_count.restype = C.c_int32
    L.rnde_node_fallback_count.argtypes = [vp]
    L.rnde_node_last_attempts.restype = C.c_int32
    L.rnde_node_last_attempts.argtypes = [vp]
    L.rnde_node_launches_per_attempt.restype = C.c_int32
    L.rnde_node_launches_per_attempt.argtypes = [vp]
    L.rnde_node_one_launch_solves.restype = C.c_int32
    L.rnde_node_one_launch_solves.argtypes = [vp]
    L.rnde_node_set_matrix_mode.restype = C.c_int32
    L.rnde_node_set_matrix_mode.argtypes = [vp, i32]
    L.rnde_node_matrix_mode.restype = C.c_int32
    L.rnde_node_matrix_mode.argtypes = [vp]
    L.rnde_classifier_head.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.rnde_node_classifier_grad.argtypes = [vp, vp, vp, vp, vp, i32, i32, f, f, f, vp, vp, vp, vp, fp, C.POINTER(C.c_int64), vp, vp]
    L.rnde_momentum_step.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, f, f, f, vp]
    u64 = C.c_uint64
    L.rnde_momentum_step_scaled.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, f, f, f, f, vp]
    L.rnde_adam_step.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    L.rnde_comm_unique_id.argtypes = [C.c_char_p]
    L.rnde_comm_create.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
    L.rnde_comm_destroy.argtypes = [vp]
    L.rnde_comm_destroy.restype = None
    L.rnde_comm_world.argtypes = [vp]
    L.rnde_comm_world.restype = i32
    L.rnde_comm_last_error.argtypes = [vp]
    L.rnde_comm_last_error.restype = C.c_char_p
    L.rnde_comm_library.argtypes = []
    L.rnde_comm_library.restype = C.c_char_p
    L.rnde_comm_allreduce.argtypes = [vp, vp, C.c_int64, i32, vp]
    L.rnde_comm_create_local_group.argtypes = [i32, i32, C.POINTER(vp)]
    L.rnde_node_set_coupling.argtypes = [vp, vp, i32]
    L.rnde_comm_health.argtypes = [vp]
    L.rnde_comm_window_create.argtypes = [i32, C.POINTER(vp), C.c_char_p]
    L.rnde_comm_window_destroy.argtypes = [vp]
    L.rnde_comm_window_destroy.restype = None
    L.rnde_comm_create_peers.argtypes = [vp, C.c_char_p, i32, i32, C.POINTER(vp)]
    L.rnde_comm_path.argtypes = [vp]
    L.rnde_comm_path.restype = C.c_char_p
    L.rnde_tapes_create.argtypes = [C.POINTER(NodeConfig), i32, C.POINTER(vp)]
    L.rnde_tapes_destroy.argtypes = [vp]
    L.rnde_tapes_destroy.restype = None
    L.rnde_tapes_last_error.argtypes = [vp]
    L.rnde_tapes_last_error.restype = C.c_char_p
    L.rnde_tapes_in_use.argtypes = [vp]
    L.rnde_tapes_node.argtypes = [vp, i32]
    L.rnde_tapes_node.restype = vp
    L.rnde_tapes_forward.argtypes = [vp, vp, vp, i32, C.c_float, C.c_float, vp, C.POINTER(C.c_int64), fp, C.POINTER(i32), i32, vp, C.POINTER(i32)]
    L.rnde_tapes_backward.argtypes = [vp, i32, vp, fp, vp, vp, fp, vp]
    L.rnde_tapes_release.argtypes = [vp, i32]
    L.rnde_nsde_param_count.restype = i32
    L.rnde_nsde_param_count.argtypes = [C.POINTER(NsdeConfig), i32p]
    L.rnde_nsde_create.argtypes = [C.POINTER(NsdeConfig), C.POINTER(vp)]
    L.rnde_nsde_destroy.argtypes = [vp]
    L.rnde_nsde_destroy.restype = None
    L.rnde_nsde_last_error.restype = C.c_char_p
    L.rnde_nsde_last_error.argtypes = [vp]
    L.rnde_nsde_forward.argtypes = [vp, vp, vp, i32, f, f, vp, i32, u64, vp, i64p, i64p, fp, i32p, i32, vp]
    L.rnde_nsde_forward_saveat.argtypes = [vp, vp, vp, i32, f, f, vp, i32, u64, fp, i32, vp, i64p, i64p, fp, i32p, i32, vp]
    L.rnde_nsde_forward_everystep.argtypes = [vp, vp, vp, i32, f, f, vp, i32, u64, i32, vp, i32, fp, i32p, i64p, i64p, fp, i32p, i32, vp]
    L.rnde_nsde_forward_replay.argtypes = [vp, vp, vp, i32, f, f, vp, i32, fp, i32, vp, i64p, i64p, fp, i32p, i32, vp]
    L.rnde_nsde_backward.argtypes = [vp, vp, fp, vp, vp, vp]
    L.rnde_nsde_backward_async.argtypes = [vp, vp, fp, vp, vp, vp]
    L.rnde_nsde_classifier_head.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.rnde_nsde_classifier_grad.argtypes = [vp, vp, vp, vp, vp, i32, i32, f, f, vp, i32, C.c_uint64, f, vp, vp, vp, vp, fp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]
    L.rnde_nsde_steps.argtypes = [vp, fp, i32, i32p, i32p]
    L.rnde_nsde_debug_attempt.argtypes = [vp, vp, vp, i32, f, vp, vp, vp, vp, fp, vp]
    L.rnde_nsde_timing.argtypes = [vp, fp, fp, i32p, i32p]
    L.rnde_normal_fill.argtypes = [vp, C.c_int64, u64, u64, vp]
    L.rnde_latent_create.argtypes = [C.POINTER(LatentConfig), C.POINTER(vp)]
    L.rnde_latent_destroy.argtypes = [vp]
    L.rnde_latent_destroy.restype = None
    L.rnde_latent_last_error.argtypes = [vp]
    L.rnde_latent_last_error.restype = C.c_char_p
    L.rnde_latent_param_counts.argtypes = [i32p, i32p, i32p]
    L.rnde_latent_param_counts.restype = None
    L.rnde_latent_encode.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.rnde_latent_decode_loss.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.rnde_latent_encode_backward.argtypes = [vp, vp, f, vp, vp, vp, vp, vp, vp]
    L.rnde_adamax_step.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, f, f, f, f, f, f, vp]
    _lib = L
    return L


EXPORTS = ["rnde_version", "rnde_last_error", "rnde_param_count", "rnde_node_create", "rnde_node_destroy",
           "rnde_node_forward", "rnde_node_forward_saveat", "rnde_node_forward_everystep", "rnde_node_forward_replay", "rnde_node_backward", "rnde_node_backward_async", "rnde_node_release_tape", "rnde_node_forward_host",
           "rnde_node_backward_host", "rnde_node_steps", "rnde_debug_feval", "rnde_debug_attempt",
           "rnde_bench_attempt", "rnde_bench_attempt_taped", "rnde_bench_attempt_cold_tape", "rnde_node_set_timing", "rnde_node_timing", "rnde_node_last_attempts", "rnde_node_fallback_count",
           "rnde_node_launches_per_attempt", "rnde_node_one_launch_solves", "rnde_node_set_matrix_mode", "rnde_node_matrix_mode", "rnde_classifier_head", "rnde_node_classifier_grad", "rnde_momentum_step", "rnde_momentum_step_scaled", "rnde_adam_step",
           "rnde_comm_unique_id", "rnde_comm_create", "rnde_comm_destroy", "rnde_comm_world", "rnde_comm_last_error", "rnde_comm_library", "rnde_comm_allreduce", "rnde_comm_create_local_group", "rnde_comm_health", "rnde_comm_window_create", "rnde_comm_window_destroy", "rnde_comm_create_peers", "rnde_comm_path", "rnde_node_set_coupling", "rnde_tapes_create", "rnde_tapes_destroy", "rnde_tapes_last_error", "rnde_tapes_in_use", "rnde_tapes_node", "rnde_tapes_forward", "rnde_tapes_backward", "rnde_tapes_release",
           "rnde_nsde_param_count", "rnde_nsde_create", "rnde_nsde_destroy", "rnde_nsde_last_error", "rnde_nsde_forward",
           "rnde_nsde_forward_saveat", "rnde_nsde_forward_everystep", "rnde_nsde_forward_replay", "rnde_nsde_backward", "rnde_nsde_backward_async", "rnde_nsde_classifier_head", "rnde_nsde_classifier_grad", "rnde_nsde_steps", "rnde_nsde_debug_attempt", "rnde_nsde_timing", "rnde_normal_fill", "rnde_latent_create", "rnde_latent_destroy", "rnde_latent_last_error", "rnde_latent_param_counts", "rnde_latent_encode",
           "rnde_latent_decode_loss", "rnde_latent_encode_backward", "rnde_adamax_step"]


def check(h, status):
    if status != OK:
        msg = lib().rnde_last_error(h).decode() if h else lib().rnde_last_error(None).decode()
        raise RndeError(status, msg)


def check_latent(h, status):
    if status != OK:
        raise RndeError(status, lib().rnde_latent_last_error(h if h else None).decode())


def check_nsde(h, status):
    if status != OK:
        raise RndeError(status, lib().rnde_nsde_last_error(h if h else None).decode())
