"""Small dense second-order-cone solver for the post-loop convex restriction
(host side, numpy; problems have a few dozen unknowns and are solved a handful of times).

    minimise  c'w   subject to   h - G w  in  R_+^p  x  Q^{q_1} x ... x Q^{q_k}

Primal-dual interior point, Nesterov-Todd scaling, Mehrotra predictor-corrector, started from a
strictly feasible ``w0``.  Same iteration as the vertex solver of the device path, written for
clarity rather than speed.
"""
from __future__ import annotations

import numpy as np


def _det(s):
    nn = np.sqrt(s[1:] @ s[1:])
    return (s[0] - nn) * (s[0] + nn)


def _nt(s, z):
    q = len(s)
    ss, zz = _det(s), _det(z)
    sb, zb = s / np.sqrt(ss), z / np.sqrt(zz)
    gam = np.sqrt((1 + zb @ sb) / 2)
    J = np.ones(q); J[1:] = -1
    wb = (sb + J * zb) / (2 * gam)
    eta = (ss / zz) ** 0.25
    W = np.empty((q, q))
    W[0, 0] = wb[0]; W[0, 1:] = wb[1:]; W[1:, 0] = wb[1:]
    W[1:, 1:] = np.eye(q - 1) + np.outer(wb[1:], wb[1:]) / (1 + wb[0])
    Wi = W.copy(); Wi[0, 1:] *= -1; Wi[1:, 0] *= -1
    return eta * W, Wi / eta


def _prod(a, b):
    return np.concatenate([[a @ b], a[0] * b[1:] + b[0] * a[1:]])


def _div(l, d):
    x0 = (l[0] * d[0] - l[1:] @ d[1:]) / _det(l)
    return np.concatenate([[x0], (d[1:] - x0 * l[1:]) / l[0]])


def _step_soc(s, ds):
    a = ds[0] ** 2 - ds[1:] @ ds[1:]
    b = 2 * (s[0] * ds[0] - s[1:] @ ds[1:])
    c = _det(s)
    al = np.inf
    if ds[0] < 0:
        al = -s[0] / ds[0]
    if abs(a) < 1e-300:
        if b < 0:
            al = min(al, -c / b)
    else:
        disc = b * b - 4 * a * c
        if disc >= 0:
            sq = np.sqrt(disc)
            qq = -0.5 * (b + (sq if b >= 0 else -sq))
            for r in (qq / a, c / qq if qq != 0 else np.inf):
                if r > 0:
                    al = min(al, r)
    return al


def solve_socp(c, G, h, p, cones, w0, tol=1e-10, max_iter=100, reg=1e-10):
    """``p`` linear rows first, then cones of the given sizes.  Returns (w, optimal value, iterations)."""
    c = np.asarray(c, float); G = np.asarray(G, float); h = np.asarray(h, float)
    nw = len(c)
    offs = np.concatenate([[p], p + np.cumsum(cones)]).astype(int)
    w = np.array(w0, float)
    s = h - G @ w
    if not (np.all(s[:p] > 0) and all(s[a] > np.linalg.norm(s[a + 1:b]) for a, b in zip(offs[:-1], offs[1:]))):
        raise ValueError("starting point is not strictly feasible")
    lam = np.empty_like(s)
    lam[:p] = 1.0 / s[:p]
    for a, b in zip(offs[:-1], offs[1:]):
        J = np.ones(b - a); J[1:] = -1
        lam[a:b] = J * s[a:b] / _det(s[a:b])
    deg = p + len(cones)
    it = 0
    for it in range(max_iter):
        s = h - G @ w
        mu = (s @ lam) / deg
        if mu <= tol:
            break
        D = np.zeros((len(s), len(s)))
        D[np.arange(p), np.arange(p)] = lam[:p] / s[:p]
        Ws = []
        for a, b in zip(offs[:-1], offs[1:]):
            W, Wi = _nt(s[a:b], lam[a:b])
            Ws.append((W, Wi, W @ lam[a:b]))
            D[a:b, a:b] = Wi @ Wi
        K = G.T @ D @ G + reg * np.eye(nw)
        rd = c + G.T @ lam

        def direction(sigmu, corr):
            t = np.zeros_like(s)
            t[:p] = (sigmu - corr[:p]) / s[:p] - lam[:p]
            for (a, b), (W, Wi, lt) in zip(zip(offs[:-1], offs[1:]), Ws):
                e = np.zeros(b - a); e[0] = 1.0
                t[a:b] = Wi @ _div(lt, sigmu * e - _prod(lt, lt) - corr[a:b])
            dw = np.linalg.solve(K, -rd - G.T @ t)
            return dw, t + D @ (G @ dw)

        def max_step(dw, dl):
            ds = -G @ dw
            al = np.inf
            for v, dv in ((s[:p], ds[:p]), (lam[:p], dl[:p])):
                neg = dv < 0
                if neg.any():
                    al = min(al, np.min(-v[neg] / dv[neg]))
            for a, b in zip(offs[:-1], offs[1:]):
                al = min(al, _step_soc(s[a:b], ds[a:b]), _step_soc(lam[a:b], dl[a:b]))
            return al

        dw, dl = direction(0.0, np.zeros_like(s))
        al = min(1.0, max_step(dw, dl))
        ds = -G @ dw
        sigma = min(1.0, max(0.0, ((s + al * ds) @ (lam + al * dl)) / deg / mu)) ** 3
        corr = ds * dl
        for (a, b), (W, Wi, lt) in zip(zip(offs[:-1], offs[1:]), Ws):
            corr[a:b] = _prod(Wi @ ds[a:b], W @ dl[a:b])
        dw, dl = direction(sigma * mu, corr)
        al = min(1.0, 0.99 * max_step(dw, dl))
        for _ in range(40):
            s2 = h - G @ (w + al * dw); l2 = lam + al * dl
            if all(s2[a] > np.linalg.norm(s2[a + 1:b]) and l2[a] > np.linalg.norm(l2[a + 1:b])
                   for a, b in zip(offs[:-1], offs[1:])) and np.all(s2[:p] > 0) and np.all(l2[:p] > 0):
                break
            al *= 0.7
        w = w + al * dw
        lam = lam + al * dl
        if al < 1e-8:
            break
    return w, float(c @ w), it
