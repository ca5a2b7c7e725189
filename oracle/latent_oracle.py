"""CPU restatement (numpy, fp64 by default) of the latent-ODE caller of the hot path -- TEST INFRASTRUCTURE, never imported by the product.

What it restates (reference experiments/latent_ode.jl, src/models/time_series.jl):
  LatentGRU / single_run        latent_ode.jl:39-106   (49 steps BACKWARDS over the time axis, three two-layer stacks, update / reset gates)
  rec_to_gen + sampling         latent_ode.jl:112, time_series.jl:50-59   (Dense(100, 50, tanh) -> Dense(50, 40); z0 = eps * exp(logvar / 2) + mu0)
  gen_to_data + losses          latent_ode.jl:148, :192-204, :226-236, time_series.jl:63-67
                                (Dense(20, 37) on every saved state; masked Gaussian log likelihood / observed count; KL to a standard normal)
and the reverse pass of each (what Tracker.gradient computes, latent_ode.jl:339-347), written out by hand.

Parity unpinned against Julia (no Julia in the image: DESIGN.md 2); pinned here by finite differences of the total loss in fp64
(tests/test_host.py::test_latent_oracle_gradients_match_finite_differences) and against the torch mirror of the same formulas
(regneuralde.jl_amd/timeseries.py).  Layouts: a Julia F x T x B array is a numpy array of shape (B, T, F); flat parameter vectors are
Flux.destructure's (each Dense as [vec(W) column-major (out x in); b]).
"""
import numpy as np

SIGMA = 0.01          # latent_ode.jl:196


def _dense_split(p, o, n_in, n_out):
    """(W as (in, out) -- out x in column-major, b, next offset)"""
    W = p[o:o + n_in * n_out].reshape(n_in, n_out)
    b = p[o + n_in * n_out:o + n_in * n_out + n_out]
    return W, b, o + n_in * n_out + n_out


def _sig(v):
    return 1.0 / (1.0 + np.exp(-v))


class GruShape:
    def __init__(self, in_dim=37, h_dim=40, latent=50):
        self.in_dim, self.h, self.L = in_dim, h_dim, latent
        self.nx = 2 * in_dim + 1
        self.n_in = 2 * latent + self.nx

    def n_params(self):
        return 2 * (self.n_in * self.h + self.h + self.h * self.L + self.L) + (self.n_in * self.h + self.h + self.h * 2 * self.L + 2 * self.L)

    def split(self, p):
        o = 0
        Wu1, bu1, o = _dense_split(p, o, self.n_in, self.h); Wu2, bu2, o = _dense_split(p, o, self.h, self.L)
        Wr1, br1, o = _dense_split(p, o, self.n_in, self.h); Wr2, br2, o = _dense_split(p, o, self.h, self.L)
        Wn1, bn1, o = _dense_split(p, o, self.n_in, self.h); Wn2, bn2, o = _dense_split(p, o, self.h, 2 * self.L)
        assert o == len(p)
        return (Wu1, bu1, Wu2, bu2, Wr1, br1, Wr2, br2, Wn1, bn1, Wn2, bn2)


def gru_forward(S, p, x):
    """x: (B, T, 2 in_dim + 1).  Returns (y (B, 2 L) = vcat(y_mean, y_std), tape)."""
    Wu1, bu1, Wu2, bu2, Wr1, br1, Wr2, br2, Wn1, bn1, Wn2, bn2 = S.split(p)
    B, T, _ = x.shape
    ym = np.zeros((B, S.L), dtype=p.dtype); ys = np.zeros((B, S.L), dtype=p.dtype)
    tape = []
    for t in range(T - 1, -1, -1):
        xt = x[:, t, :]
        yc = np.concatenate([ym, ys, xt], axis=1)
        U1 = np.tanh(yc @ Wu1 + bu1); u = _sig(U1 @ Wu2 + bu2)
        R1 = np.tanh(yc @ Wr1 + br1); r = _sig(R1 @ Wr2 + br2)
        cc = np.concatenate([ym * r, ys * r, xt], axis=1)
        N1 = np.tanh(cc @ Wn1 + bn1); ns = N1 @ Wn2 + bn2
        nsm, nss = ns[:, :S.L], ns[:, S.L:]
        m = (xt[:, S.nx // 2:].sum(axis=1, keepdims=True) > 0).astype(p.dtype)      # rows (size / 2 + 1):end: the mask rows and the time row
        nym = m * ((1 - u) * nsm + u * ym) + (1 - m) * ym
        nys = m * ((1 - u) * nss + u * ys) + (1 - m) * ys
        tape.append((t, ym, ys, yc, U1, u, R1, r, cc, N1, nsm, nss, m))
        ym, ys = nym, nys
    return np.concatenate([ym, ys], axis=1), tape


def gru_backward(S, p, tape, ybar):
    """ybar: (B, 2 L).  Returns p-bar (flat, Flux.destructure order)."""
    Wu1, bu1, Wu2, bu2, Wr1, br1, Wr2, br2, Wn1, bn1, Wn2, bn2 = S.split(p)
    g = [np.zeros_like(a) for a in (Wu1, bu1, Wu2, bu2, Wr1, br1, Wr2, br2, Wn1, bn1, Wn2, bn2)]
    ymb, ysb = ybar[:, :S.L].copy(), ybar[:, S.L:].copy()
    for (t, ym, ys, yc, U1, u, R1, r, cc, N1, nsm, nss, m) in reversed(tape):
        gm, gs = m * ymb, m * ysb                      # cotangents of the gated update
        ymb_o, ysb_o = (1 - m) * ymb + u * gm, (1 - m) * ysb + u * gs
        ub = (ym - nsm) * gm + (ys - nss) * gs
        nsb = np.concatenate([(1 - u) * gm, (1 - u) * gs], axis=1)
        g[10] += N1.T @ nsb; g[11] += nsb.sum(0)
        zb = (nsb @ Wn2.T) * (1 - N1 * N1)
        g[8] += cc.T @ zb; g[9] += zb.sum(0)
        ccb = zb @ Wn1.T
        rb = ym * ccb[:, :S.L] + ys * ccb[:, S.L:2 * S.L]
        ymb_o += r * ccb[:, :S.L]; ysb_o += r * ccb[:, S.L:2 * S.L]
        au = ub * u * (1 - u); ar = rb * r * (1 - r)
        g[2] += U1.T @ au; g[3] += au.sum(0); g[6] += R1.T @ ar; g[7] += ar.sum(0)
        zu = (au @ Wu2.T) * (1 - U1 * U1); zr = (ar @ Wr2.T) * (1 - R1 * R1)
        g[0] += yc.T @ zu; g[1] += zu.sum(0); g[4] += yc.T @ zr; g[5] += zr.sum(0)
        ycb = zu @ Wu1.T + zr @ Wr1.T
        ymb, ysb = ymb_o + ycb[:, :S.L], ysb_o + ycb[:, S.L:2 * S.L]
    return np.concatenate([a.reshape(-1) for a in g])


def encode_forward(p2, y, eps, rec=50, latent=20):
    """rec_to_gen + the reparameterised sample.  y: (B, 2 rec), eps: (B, latent).  Returns (z0, mu0, logvar, tape)."""
    W1, b1, o = _dense_split(p2, 0, y.shape[1], rec); W2, b2, o = _dense_split(p2, o, rec, 2 * latent)
    h = np.tanh(y @ W1 + b1)
    out = h @ W2 + b2
    mu0, lv = out[:, :latent], out[:, latent:]
    z0 = eps * np.exp(lv / 2) + mu0
    return z0, mu0, lv, (y, h, mu0, lv, eps)


def encode_backward(p2, tape, z0bar, kl_weight, rec=50, latent=20):
    """z0bar: cotangent from the solve; kl_weight = lambda_k / B: the loss holds + kl_weight * sum_b KL_b, KL_b = mean_i(exp(lv) + mu^2 - 1 - lv) / 2.
    Returns (ybar (B, 2 rec), p2bar)."""
    y, h, mu0, lv, eps = tape
    W1, b1, o = _dense_split(p2, 0, y.shape[1], rec); W2, b2, o = _dense_split(p2, o, rec, 2 * latent)
    mub = z0bar + kl_weight * mu0 / latent
    lvb = z0bar * eps * np.exp(lv / 2) / 2 + kl_weight * (np.exp(lv) - 1) / (2 * latent)
    ob = np.concatenate([mub, lvb], axis=1)
    gW2, gb2 = h.T @ ob, ob.sum(0)
    zb = (ob @ W2.T) * (1 - h * h)
    gW1, gb1 = y.T @ zb, zb.sum(0)
    return zb @ W1.T, np.concatenate([gW1.reshape(-1), gb1, gW2.reshape(-1), gb2])


def kl_per_sample(mu0, lv):
    return (np.exp(lv) + mu0 * mu0 - 1 - lv).mean(axis=1) / 2


def decode_loss(p4, res, data, mask):
    """gen_to_data on every saved state + the masked Gaussian log likelihood.  res: (B, T, latent), data / mask: (B, T, in_dim).
    Returns (nll = -mean_b ll_b, res-bar, p4-bar) for the loss term -mean_b ll_b (the reference counts the constants at unobserved entries too)."""
    B, T, Lz = res.shape
    W, b, _ = _dense_split(p4, 0, Lz, data.shape[2])
    pred = res.reshape(B * T, Lz) @ W + b
    d = (pred.reshape(B, T, -1) * mask - data * mask)
    M = mask.sum(axis=(1, 2))
    ll = (-(d * d) / (2 * SIGMA ** 2) - np.log(SIGMA) - np.log(2 * np.pi) / 2).sum(axis=(1, 2)) / M
    nll = -ll.mean()
    predb = (d * mask / (SIGMA ** 2)) / (M[:, None, None] * B)
    pb = predb.reshape(B * T, -1)
    return nll, (pb @ W.T).reshape(B, T, Lz), np.concatenate([(res.reshape(B * T, Lz).T @ pb).reshape(-1), pb.sum(0)]), ll
