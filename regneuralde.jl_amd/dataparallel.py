"""Data parallelism for the training step: one process per GPU, minibatch columns sharded across ranks,
ONE sum-all-reduce of the flat gradient per step (RCCL over xGMI when the backend is "nccl").

The reference is single-process (no collectives); the only place a collective belongs is
between Tracker.gradient and update_parameters! (reference experiments/mnist_node.jl:229-233,
src/utils.jl:149-156).  Each rank integrates its own shard with its own step-size controller
(mode 1): no data-path collective exists.  The payload for MNIST-NODE is 166,418 fp32
(665,672 B): latency-bound, so it is sent as ONE contiguous buffer.
"""
import torch


class GradientAllReducer:
    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.numel() > 0]
        self.pg = process_group
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)

    @torch.no_grad()
    def allreduce_(self):
        """Average .grad of every parameter group over the ranks, in place."""
        import torch.distributed as dist
        world = dist.get_world_size(self.pg)
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, group=self.pg)
        self.flat.div_(world)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad.copy_(self.flat[off:off + n].view_as(p.grad))
            off += n


def shard_columns(x, rank, world):
    """Contiguous column blocks: rank g holds samples [g*B/G, (g+1)*B/G)."""
    B = x.shape[0]
    per = (B + world - 1) // world
    return x[rank * per:min(B, (rank + 1) * per)]
