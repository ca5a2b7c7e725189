"""world_size = 2 on CPU (gloo): the N > 1 path of the training step -- flat gradient all-reduce and
identical optimiser updates on every rank."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, store_path, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import regneuralde_jl_amd as rn
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    # rendezvous through a file (no fixed TCP port to collide with a socket in TIME_WAIT or another run on this host)
    dist.init_process_group("gloo", init_method=f"file://{store_path}", rank=rank, world_size=world)
    torch.manual_seed(0)
    p1 = torch.zeros(0)
    p2 = torch.randn(1000, requires_grad=True)
    p3 = torch.randn(37, requires_grad=True)
    g = torch.Generator().manual_seed(100 + rank)
    p2.grad = torch.randn(1000, generator=g)
    p3.grad = torch.randn(37, generator=g)
    local = torch.cat([p2.grad, p3.grad]).clone()
    red = rn.GradientAllReducer([p1, p2, p3])
    red.allreduce_()
    opt = rn.FluxOptimiser([p1, p2, p3])
    avg = torch.cat([p2.grad, p3.grad]).clone()
    opt.step()
    # (numpy, pickled by value: a tensor would travel as a file descriptor of this process, which may be gone by the time the parent reads the queue)
    q.put((rank, local.numpy(), avg.numpy(), torch.cat([p2.detach(), p3.detach()]).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_and_update_world2():
    import tempfile
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    fd, store_path = tempfile.mkstemp(prefix="rnde_gloo_")
    os.close(fd)
    os.unlink(store_path)                      # the FileStore creates it
    procs = [ctx.Process(target=_worker, args=(r, 2, store_path, q)) for r in range(2)]
    try:
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
        if os.path.exists(store_path):
            os.unlink(store_path)
    (_, l0, a0, w0), (_, l1, a1, w1) = [(r, *(torch.from_numpy(a) for a in rest)) for r, *rest in res]
    assert torch.allclose(a0, (l0 + l1) / 2, atol=1e-6) and torch.equal(a0, a1)   # mean of the rank gradients
    assert torch.equal(w0, w1)                                                     # replicas stay identical
