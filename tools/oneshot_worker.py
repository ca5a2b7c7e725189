"""One rank of the one-shot all-reduce (include/rnde.h: rnde_comm_window_create / rnde_comm_create_peers) as its own PROCESS.

  python tools/oneshot_worker.py --rank R --world W --dir DIR [--device D] [--rounds K] [--sizes 1,5,1023,166418,262144,300000]

Ranks meet through files in DIR (the 64-byte window handles: h<rank>.bin), build the peer communicator, then all-reduce `rounds`
seeded vectors per size -- every rank can generate every rank's input, so each checks its own result BIT FOR BIT against the
rank-order fp32 sum -- and time 200 back-to-back all-reduces of the MNIST-NODE gradient buffer (166,418 floats) with HIP events.
Prints one JSON line.  tests/test_gpu_comm.py starts W of these on the one GPU of the test box (device 0 for every rank: peers
mapped through hipIpc on the same device); on a multi-GPU node pass --device R."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--sizes", default="1,5,1023,4097,166418,262144,300001")
    ap.add_argument("--mean", type=int, default=0)
    ap.add_argument("--absent", type=int, default=-1, help="this rank builds the communicator and then makes NO all-reduce: the others must time out and say so")
    a = ap.parse_args()
    import torch
    from regneuralde_jl_amd import _lib
    L = _lib.lib()
    torch.cuda.set_device(a.device)
    win, h = C.c_void_p(), C.create_string_buffer(64)
    st = L.rnde_comm_window_create(a.device, C.byref(win), h)
    assert st == 0, L.rnde_comm_last_error(None)
    tmp = os.path.join(a.dir, "h%d.tmp" % a.rank)
    open(tmp, "wb").write(h.raw)
    os.rename(tmp, os.path.join(a.dir, "h%d.bin" % a.rank))
    handles, t0 = b"", time.time()
    for r in range(a.world):
        f = os.path.join(a.dir, "h%d.bin" % r)
        while not os.path.exists(f):
            assert time.time() - t0 < 120, "rank %d never exported its window" % r
            time.sleep(0.01)
        handles += open(f, "rb").read()
    comm = C.c_void_p()
    st = L.rnde_comm_create_peers(win, handles, a.rank, a.world, C.byref(comm))
    assert st == 0, L.rnde_comm_last_error(None)
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    if a.absent >= 0:   # failure drill: one rank stays away from the first all-reduce
        if a.rank != a.absent:
            g = torch.ones(1000, device="cuda")
            st = L.rnde_comm_allreduce(comm, g.data_ptr(), 1000, 0, sp)
            stream.synchronize()
            # what a training loop meets WITHOUT calling rnde_comm_health: the reduced buffer is poisoned, the next enqueue fails (sticky), on any stream
            nan_frac = float(torch.isnan(g).float().mean())
            g2 = torch.ones(1000, device="cuda")
            st2 = L.rnde_comm_allreduce(comm, g2.data_ptr(), 1000, 0, None)
            err2 = L.rnde_comm_last_error(comm).decode()
            health = L.rnde_comm_health(comm)
            print(json.dumps({"rank": a.rank, "enqueue": st, "nan_frac": nan_frac, "next_enqueue": st2, "next_error": err2, "health": health,
                              "error": L.rnde_comm_last_error(comm).decode()}), flush=True)
        else:
            time.sleep(3.0)      # keep the window mapped while the others wait
            print(json.dumps({"rank": a.rank, "absent": True}), flush=True)
        L.rnde_comm_destroy(comm)
        return
    bad, checked = 0, 0
    with torch.cuda.stream(stream):
        for n in [int(x) for x in a.sizes.split(",")]:
            for k in range(a.rounds):
                xs = [np.random.default_rng([n, k, r]).standard_normal(n + 1).astype(np.float32) for r in range(a.world)]
                off = k & 1                                             # odd rounds: a buffer that is NOT 16-byte aligned
                buf = torch.from_numpy(xs[a.rank]).cuda()
                st = L.rnde_comm_allreduce(comm, buf.data_ptr() + 4 * off, n, a.mean, sp)
                assert st == 0, L.rnde_comm_last_error(comm)
                stream.synchronize()
                got = buf.cpu().numpy()
                full = xs[0][off:off + n].copy()                        # the reduced range, summed in rank order in fp32
                for r in range(1, a.world):
                    full = full + xs[r][off:off + n]
                want = xs[a.rank].copy()
                want[off:off + n] = full * np.float32(1.0 / a.world) if a.mean else full
                bad += int(not np.array_equal(got, want))
                checked += 1
        # latency of the gradient message
        n = 166418
        g = torch.randn(n, device="cuda")
        for _ in range(20):
            L.rnde_comm_allreduce(comm, g.data_ptr(), n, 0, sp)
            g.mul_(0.1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(200):
            L.rnde_comm_allreduce(comm, g.data_ptr(), n, 1, sp)
        e1.record(stream)
        stream.synchronize()
    health = L.rnde_comm_health(comm)
    print(json.dumps({"rank": a.rank, "world": a.world, "checked": checked, "mismatches": bad, "health": health,
                      "allreduce_us_166418": round(e0.elapsed_time(e1) * 1000 / 200, 2), "path": L.rnde_comm_path(comm).decode()}), flush=True)
    L.rnde_comm_destroy(comm)


if __name__ == "__main__":
    main()
