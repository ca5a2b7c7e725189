"""Data parallelism for the training step: one process per GPU, minibatch columns sharded across ranks,
ONE sum-all-reduce of the flat gradient per step.

The reference is single-process (no collectives); the only place a collective belongs is
between Tracker.gradient and update_parameters! (reference experiments/mnist_node.jl:229-233,
src/utils.jl:149-156).  Each rank integrates its own shard with its own step-size controller
(mode 1): no data-path collective exists.  The payload for MNIST-NODE is 166,418 fp32
(665,672 B): latency-bound, so it lives in ONE contiguous buffer that the reverse pass writes directly
(`FlatGrads`: p.grad are views of it -- no gather / scatter copies around the collective).

On cuda tensors the collective is the library's own (include/rnde.h: rnde_comm_*, RCCL over xGMI on the
caller's stream; the unique id travels through the torch.distributed store); this class is a thin caller
of it, exactly what a Julia caller would write.  CPU tensors (the gloo tests) use torch.distributed.
The averaging (1 / world) is folded into the optimiser launch (`grad_scale` -> rnde_momentum_step_scaled).
"""
import ctypes as C

import torch


class FlatGrads:
    """One contiguous gradient buffer for a list of parameter groups; `views[i]` is group i's slice (use as p.grad / as the
    output pointer of the reverse pass)."""

    def __init__(self, params):
        self.params = [p for p in params if p.numel() > 0]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()


class GradientAllReducer:
    """collective: "rccl" (default; RNDE_ONESHOT=1 in the environment adds the one-shot kernel for buffers up to 262,144 floats, its
    window handles carried by RCCL itself) or "peers" (no RCCL at all: rnde_comm_create_peers, the window handles travel through
    torch.distributed's object all-gather -- any backend, e.g. gloo; also what lets several ranks share ONE GPU in the tests).
    Default from RNDE_COLLECTIVE."""

    def __init__(self, params, process_group=None, flat=None, collective=None):
        import os
        import torch.distributed as dist
        self.collective = collective or os.environ.get("RNDE_COLLECTIVE", "rccl")
        if self.collective not in ("rccl", "peers"):
            raise ValueError("collective: 'rccl' or 'peers'")
        self.params = [p for p in params if p.numel() > 0]
        self.pg = process_group
        self.world = dist.get_world_size(self.pg)
        self.fg = flat if flat is not None else FlatGrads(self.params)
        self.flat = self.fg.flat
        self.comm = None
        self.fallback_reason = None
        if self.flat.is_cuda:
            self._init_comm(dist)

    # -- the library's communicator: rank 0 makes the id, the torch.distributed store carries it ---------------------------
    def _init_comm(self, dist):
        from . import _lib
        L = _lib.lib()
        rank = dist.get_rank(self.pg)
        self._L = L
        if self.collective == "peers":
            dev = self.flat.device.index or 0
            win, h = C.c_void_p(), C.create_string_buffer(64)
            st = L.rnde_comm_window_create(dev, C.byref(win), h)
            if st != 0:
                raise _lib.RndeError(st, L.rnde_comm_last_error(None).decode())
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(h.raw), group=self.pg)
            self.comm = C.c_void_p()
            st = L.rnde_comm_create_peers(win, b"".join(handles), rank, self.world, C.byref(self.comm))
            if st != 0:
                L.rnde_comm_window_destroy(win)
                raise _lib.RndeError(st, L.rnde_comm_last_error(None).decode())
            return
        ids = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            st = L.rnde_comm_unique_id(buf)
            if st != 0:
                raise _lib.RndeError(st, L.rnde_comm_last_error(None).decode())
            ids = [bytes(buf.raw)]
        dist.broadcast_object_list(ids, src=0, group=self.pg)
        self.comm = C.c_void_p()
        st = L.rnde_comm_create(ids[0], rank, self.world, self.flat.device.index or 0, C.byref(self.comm))
        self._L = L
        reason = None if st == 0 else L.rnde_comm_last_error(None).decode()
        if self.world > 1:
            # every rank must end up on the SAME path: if the library's communicator failed anywhere (it has never met more than one GPU: no such
            # box exists where it was written), all ranks drop it and the gradient all-reduce goes through torch.distributed's RCCL instead --
            # slower by a launch or two, never wrong, and reported (`fallback_reason`, the bench's `dist.collective_fallback`)
            ok = torch.tensor([1 if st == 0 else 0], dtype=torch.int32, device=self.flat.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.pg)
            if int(ok.item()) == 0:
                if st == 0:
                    L.rnde_comm_destroy(self.comm)
                    reason = "another rank could not create its library communicator"
                self.comm = None
                self.fallback_reason = reason
                import sys
                print(f"[rnde] rank {rank}: library communicator unavailable ({reason}); gradient all-reduce through torch.distributed", file=sys.stderr)
                return
        elif st != 0:
            raise _lib.RndeError(st, reason)

    def __del__(self):
        try:
            if self.comm:
                self._L.rnde_comm_destroy(self.comm)
                self.comm = None
        except Exception:
            pass

    @property
    def grad_scale(self):
        """What the optimiser multiplies the summed gradient by (InvDecay/Momentum launch: rnde_momentum_step_scaled)."""
        return 1.0 / self.world

    def _gather(self):
        """Bring p.grad into the flat buffer unless it already lives there (FlatGrads views)."""
        for p, v in zip(self.params, self.fg.views):
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad.reshape(v.shape))
                p.grad = v

    @torch.no_grad()
    def allreduce_range_(self, lo, hi, mean=False):
        """Sum flat[lo:hi] over the ranks, asynchronously on the current stream (cuda) -- e.g. the head's gradient while the
        reverse sweep of the solve is still running."""
        seg = self.flat[lo:hi]
        if self.comm is not None:
            st = self._L.rnde_comm_allreduce(self.comm, seg.data_ptr(), seg.numel(), 1 if mean else 0,
                                             C.c_void_p(torch.cuda.current_stream(seg.device).cuda_stream))
            if st != 0:
                from . import _lib
                raise _lib.RndeError(st, self._L.rnde_comm_last_error(self.comm).decode())
        else:
            import torch.distributed as dist
            dist.all_reduce(seg, group=self.pg)
            if mean:
                seg.div_(self.world)

    @torch.no_grad()
    def allreduce_(self, mean=True):
        """Sum (mean=True: average) .grad of every parameter group over the ranks, in place, ONE collective."""
        self._gather()
        self.allreduce_range_(0, self.flat.numel(), mean)
        for p, v in zip(self.params, self.fg.views):
            p.grad = v


def shard_columns(x, rank, world, equal=False):
    """Contiguous column blocks: rank g holds samples [g*B/G, (g+1)*B/G).  equal=True (required with the coupled
    controller, `TrackedNeuralODE.set_coupling`: its per-step all-reduce adds per-workgroup arrays element-wise) refuses a batch that
    does not divide by the number of ranks instead of giving the last rank a shorter shard."""
    B = x.shape[0]
    if equal and B % world != 0:
        raise ValueError(f"shard_columns: batch {B} does not split into {world} equal shards (coupled controller needs equal shards)")
    per = (B + world - 1) // world
    return x[rank * per:min(B, (rank + 1) * per)]
