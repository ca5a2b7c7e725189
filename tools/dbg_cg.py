import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
import regneuralde_jl_amd as rn
def make(tol):
    g = torch.Generator().manual_seed(4)
    dyn = rn.MLPDynamics(784, 100, generator=g)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", reltol=tol, abstol=tol, max_batch=32, max_attempts=64)
    m = rn.ClassifierNODE(node, rn.Dense(784, 10, generator=g), device=torch.device("cuda", 0))
    x = torch.rand(32, 1, 28, 28, generator=g).cuda()
    y = torch.eye(10)[torch.randint(0, 10, (32,), generator=g)].cuda()
    return m, x, y
which = sys.argv[1]
if which == "a":    # one-call first on a fresh model, loose tolerance (few attempts: one chunk)
    m, x, y = make(1e-1)
    print(rn.fused_loss_and_grad(m, x, y, sync=False)); torch.cuda.synchronize(); print("a ok")
if which == "b":    # warm the head with the three calls, then a tolerance that needs > 4 attempts with predicted reset (new model shares nothing)
    m, x, y = make(1e-3)
    print(rn.fused_loss_and_grad(m, x, y, sync=False)); torch.cuda.synchronize(); print("b ok")
if which == "c":    # kernels loaded by ANOTHER model's three-call step; this model's own head workspace still unallocated
    m0, x0, y0 = make(1e-1)
    rn.fused_loss_and_grad(m0, x0, y0, sync=True)
    m, x, y = make(1e-1)
    print(rn.fused_loss_and_grad(m, x, y, sync=False)); torch.cuda.synchronize(); print("c ok")
if which == "d":    # non-default stream
    m, x, y = make(1e-1)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        print(rn.fused_loss_and_grad(m, x, y, sync=False))
    torch.cuda.synchronize(); print("d ok")
