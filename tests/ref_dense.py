"""Test-only cross-check: the v3 vertex sub-problem in the reference's FULL form
(every variable and constraint of admm_solver_v3.py:352-466, nothing reduced),
solved by a small dense primal-dual interior-point method written with numpy,
and the v3 outer loop (admm_solver_v3.py:543-733) around it.

Purpose: an independent formulation against which the C oracle (reduced "arrow"
form, oracle/gcs_oracle.c) is checked on small cases; it is slow (dense KKT per
vertex per iteration) and is never shipped, timed or used by the product.

The s / t vertices are points (boxes of half-width 1e-6, utils.py:12-28): their
sub-problems are solved in closed form exactly as DESIGN.md section 3 describes
(simplex projection), which is also what the oracle and the HIP path do.
"""
from __future__ import annotations

import numpy as np


# ----------------------------------------------------------------------------
# generic dense conic QP:  min 1/2 w'Hw + c'w  s.t. Ew = f,  Gw <= h (first p rows),
# h_soc - G_soc w in SOC (last q rows).  Starts from a w0 strictly inside the cones.
# ----------------------------------------------------------------------------
def _soc_scaling(s, z):
    J = np.ones_like(s); J[1:] = -1
    ss = s[0] ** 2 - s[1:] @ s[1:]
    zz = z[0] ** 2 - z[1:] @ z[1:]
    sb = s / np.sqrt(ss); zb = z / np.sqrt(zz)
    gam = np.sqrt((1 + zb @ sb) / 2)
    wb = (sb + J * zb) / (2 * gam)
    eta = (ss / zz) ** 0.25
    q = len(s)
    W = np.zeros((q, q))
    W[0, 0] = wb[0]; W[0, 1:] = wb[1:]; W[1:, 0] = wb[1:]
    W[1:, 1:] = np.eye(q - 1) + np.outer(wb[1:], wb[1:]) / (1 + wb[0])
    Winv = W.copy(); Winv[0, 1:] *= -1; Winv[1:, 0] *= -1
    return eta * W, Winv / eta


def _soc_prod(a, b):
    return np.concatenate([[a @ b], a[0] * b[1:] + b[0] * a[1:]])


def _soc_div(l, d):
    """x with l o x = d"""
    det = l[0] ** 2 - l[1:] @ l[1:]
    x0 = (l[0] * d[0] - l[1:] @ d[1:]) / det
    x1 = (d[1:] - x0 * l[1:]) / l[0]
    return np.concatenate([[x0], x1])


def _soc_max_step(s, ds):
    a = ds[0] ** 2 - ds[1:] @ ds[1:]
    b = 2 * (s[0] * ds[0] - s[1:] @ ds[1:])
    c = s[0] ** 2 - s[1:] @ s[1:]
    al = np.inf
    if ds[0] < 0:
        al = min(al, -s[0] / ds[0])
    # smallest positive root of a t^2 + b t + c = 0 (c > 0)
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


def conic_qp(H, c, E, f, G, h, q, w0, tol=1e-11, maxit=80, reg=1e-12):
    nw = len(c); p = G.shape[0] - q
    H = H + reg * np.eye(nw)     # tiny curvature: some full-form variables appear nowhere
    w = w0.copy()
    s = h - G @ w
    assert np.all(s[:p] > 0) and (q == 0 or s[p] > np.linalg.norm(s[p + 1:])), "w0 not interior"
    mu0 = 1.0
    lam = np.empty(p + q)
    lam[:p] = mu0 / s[:p]
    if q:
        ss = s[p] ** 2 - s[p + 1:] @ s[p + 1:]
        lam[p:] = mu0 * np.concatenate([[s[p]], -s[p + 1:]]) / ss
    nu = np.zeros(E.shape[0])
    deg = p + (1 if q else 0)
    for it in range(maxit):
        s = h - G @ w
        rd = H @ w + c + G.T @ lam + E.T @ nu
        rp = E @ w - f
        mu = (s @ lam) / deg
        scale = 1 + np.abs(c).max()
        if mu <= tol and (mu <= 1e-3 * tol or (np.abs(rd).max() <= 1e-7 * scale and np.abs(rp).max() <= 1e-9)):
            break
        D = np.zeros((p + q, p + q))
        D[np.arange(p), np.arange(p)] = lam[:p] / s[:p]
        if q:
            W, Winv = _soc_scaling(s[p:], lam[p:])
            D[p:, p:] = Winv @ Winv
            lt = W @ lam[p:]
        K = H + G.T @ D @ G
        KKT = np.block([[K, E.T], [E, np.zeros((E.shape[0],) * 2)]])
        KKT[nw:, nw:] -= 1e-14 * np.eye(E.shape[0])

        def solve(sigmu, corr_lp, corr_soc):
            # returns dw, dnu, dlam
            t = np.zeros(p + q)          # the vector added as G' t on the rhs
            t[:p] = (sigmu - corr_lp) / s[:p] - lam[:p]
            if q:
                rc = sigmu * np.eye(q)[0] - _soc_prod(lt, lt) - corr_soc
                qv = _soc_div(lt, rc)
                t[p:] = Winv @ qv
            rhs = np.concatenate([-rd - G.T @ t, -rp])
            sol = np.linalg.solve(KKT, rhs)
            dw, dnu = sol[:nw], sol[nw:]
            dlam = t + D @ (G @ dw)
            return dw, dnu, dlam

        def max_step(dw, dlam):
            ds = -G @ dw
            al = 1e30
            neg = ds[:p] < 0
            if neg.any():
                al = min(al, np.min(-s[:p][neg] / ds[:p][neg]))
            neg = dlam[:p] < 0
            if neg.any():
                al = min(al, np.min(-lam[:p][neg] / dlam[:p][neg]))
            if q:
                al = min(al, _soc_max_step(s[p:], ds[p:]), _soc_max_step(lam[p:], dlam[p:]))
            return al

        dw, dnu, dlam = solve(0.0, np.zeros(p), np.zeros(q))
        al = min(1.0, max_step(dw, dlam))
        ds = -G @ dw
        mu_aff = ((s + al * ds) @ (lam + al * dlam)) / deg
        sigma = min(1.0, max(0.0, mu_aff / mu)) ** 3
        corr_lp = ds[:p] * dlam[:p]
        corr_soc = _soc_prod(Winv @ ds[p:], W @ dlam[p:]) if q else np.zeros(0)
        dw, dnu, dlam = solve(sigma * mu, corr_lp, corr_soc)
        al = min(1.0, 0.99 * max_step(dw, dlam))
        if q:   # guard the cone against round-off in the root computation
            for _ in range(30):
                s2 = h[p:] - G[p:] @ (w + al * dw); l2 = lam[p:] + al * dlam[p:]
                if s2[0] > np.linalg.norm(s2[1:]) and l2[0] > np.linalg.norm(l2[1:]):
                    break
                al *= 0.7
        w = w + al * dw; nu = nu + al * dnu; lam = lam + al * dlam
    return w, it


# ----------------------------------------------------------------------------
# simplex projection used by the closed-form s / t sub-problems
# ----------------------------------------------------------------------------
def project_simplex(v):
    u = np.sort(v)[::-1]
    css = np.cumsum(u)
    k = np.nonzero(u * np.arange(1, len(v) + 1) > (css - 1))[0][-1]
    tau = (css[k] - 1) / (k + 1.0)
    return np.maximum(v - tau, 0)


# ----------------------------------------------------------------------------
# one vertex sub-problem in the reference's full form
# ----------------------------------------------------------------------------
def solve_vertex_full(n, A, b, p_int, inc, rho, eps_edge=1e-4):
    """inc: list of dicts {out: bool, T_u: target of z_{e,u}[:n], T_w: target of
    z_{e,w}[:n], T_y: target of y_e}  for e=(u,w), in I_in + I_out order.
    Returns (x_v, z_v, y_v, list of (zu_copy[:n], zw_copy[:n], y_copy))."""
    N2 = 2 * n; d = len(inc); m = A.shape[0]
    # variable layout: x_v (N2) | z_v (N2) | y_v | per e: P_e (N2), Q_e (N2), y_e | t
    ix = 0; iz = N2; iy = 2 * N2; ib = 2 * N2 + 1
    bw = 2 * N2 + 1
    nw = ib + d * bw + 1
    it_ = nw - 1
    P = lambda k: ib + k * bw
    Q = lambda k: ib + k * bw + N2
    Y = lambda k: ib + k * bw + 2 * N2
    own = lambda k: P(k) if inc[k]["out"] else Q(k)
    H = np.zeros((nw, nw)); c = np.zeros(nw)
    c[it_] = 1.0
    for k, e in enumerate(inc):
        c[Y(k)] += eps_edge
        for dd in range(n):
            H[P(k) + dd, P(k) + dd] += rho; c[P(k) + dd] -= rho * e["T_u"][dd]
            H[Q(k) + dd, Q(k) + dd] += rho; c[Q(k) + dd] -= rho * e["T_w"][dd]
        H[Y(k), Y(k)] += rho; c[Y(k)] -= rho * e["T_y"]
    rows = []; rh = []

    def add(coefs, rhs):
        r = np.zeros(nw)
        for i, v in coefs:
            r[i] += v
        rows.append(r); rh.append(rhs)
    for i in range(2):
        for j in range(m):
            add([(iz + i * n + k, A[j, k]) for k in range(n)] + [(iy, -b[j])], 0.0)            # 1
            add([(ix + i * n + k, A[j, k]) for k in range(n)] + [(iz + i * n + k, -A[j, k]) for k in range(n)]
                + [(iy, b[j])], b[j])                                                           # 2
    add([(iy, -1.0)], 0.0); add([(iy, 1.0)], 1.0)
    for k in range(d):
        o = own(k)
        for i in range(2):
            for j in range(m):
                add([(o + i * n + kk, A[j, kk]) for kk in range(n)] + [(Y(k), -b[j])], 0.0)     # 3
                add([(ix + i * n + kk, A[j, kk]) for kk in range(n)] + [(o + i * n + kk, -A[j, kk]) for kk in range(n)]
                    + [(Y(k), b[j])], b[j])                                                     # 4
        add([(Y(k), -1.0)], 0.0); add([(Y(k), 1.0)], 1.0)
    # SOC rows: h - G w = (t, z1 - z2)
    add([(it_, -1.0)], 0.0)
    for k in range(n):
        add([(iz + k, -1.0), (iz + n + k, 1.0)], 0.0)
    G = np.array(rows); h = np.array(rh)
    er = []; ef = []

    def eq(coefs, rhs):
        r = np.zeros(nw)
        for i, v in coefs:
            r[i] += v
        er.append(r); ef.append(rhs)
    for k in range(d):
        for dd in range(n):
            eq([(P(k) + n + dd, 1.0), (Q(k) + dd, -1.0)], 0.0)                                  # 5
    ins = [k for k in range(d) if not inc[k]["out"]]; outs = [k for k in range(d) if inc[k]["out"]]
    eq([(iy, 1.0)] + [(Y(k), -1.0) for k in ins], 0.0)                                          # 6
    eq([(iy, 1.0)] + [(Y(k), -1.0) for k in outs], 0.0)
    for dd in range(N2):
        eq([(iz + dd, 1.0)] + [(own(k) + dd, -1.0) for k in ins], 0.0)                           # 7
        eq([(iz + dd, 1.0)] + [(own(k) + dd, -1.0) for k in outs], 0.0)
    E = np.array(er); f = np.array(ef)
    # strictly feasible start
    w0 = np.zeros(nw); pp = np.concatenate([p_int, p_int]); c0 = 0.5
    w0[ix:ix + N2] = pp; w0[iz:iz + N2] = c0 * pp; w0[iy] = c0; w0[it_] = 1.0
    for k in range(d):
        ye = c0 / (len(outs) if inc[k]["out"] else len(ins))
        w0[Y(k)] = ye
        w0[own(k):own(k) + N2] = ye * pp
        if inc[k]["out"]:
            w0[Q(k):Q(k) + n] = w0[P(k) + n:P(k) + N2]
        else:
            w0[P(k) + n:P(k) + N2] = w0[Q(k):Q(k) + n]
    w, its = conic_qp(H, c, E, f, G, h, n + 1, w0, reg=1e-9)
    copies = [(w[P(k):P(k) + n].copy(), w[Q(k):Q(k) + n].copy(), w[Y(k)]) for k in range(d)]
    return w[ix:ix + N2].copy(), w[iz:iz + N2].copy(), w[iy], copies, its


def solve_terminal(n, pt, inc, rho, is_source, eps_edge=1e-4):
    """closed form for the point vertices s (is_source) / t."""
    d = len(inc)
    act = [k for k in range(d) if inc[k]["out"] == is_source]
    copies = [None] * d
    if act:
        if is_source:   # out-edges e=(s,b): both z halves equal y*pt
            a = 2 * pt @ pt + 1
            cc = np.array([pt @ inc[k]["T_u"] + pt @ inc[k]["T_w"] + inc[k]["T_y"] for k in act])
        else:           # in-edges e=(a,t): only z_{e,t}[:n] and y are penalised
            a = pt @ pt + 1
            cc = np.array([pt @ inc[k]["T_w"] + inc[k]["T_y"] for k in act])
        y = project_simplex((cc - eps_edge / rho) / a)
        for yy, k in zip(y, act):
            if is_source:
                copies[k] = (yy * pt, yy * pt, yy)
            else:
                copies[k] = (np.array(inc[k]["T_u"], float), yy * pt, yy)
    for k in range(d):
        if copies[k] is None:
            # the dead side: y = 0, own z = 0, the foreign coupled word is free -> its target
            if inc[k]["out"]:       # e=(t,b) at t: own = z_{e,t} (u side); foreign z_{e,b}[:n] = own second half = 0
                copies[k] = (np.zeros(n), np.zeros(n), 0.0)
            else:                   # e=(a,s) at s: foreign z_{e,a}[:n] only penalised -> target
                copies[k] = (np.array(inc[k]["T_u"], float), np.zeros(n), 0.0)
    pp = np.concatenate([pt, pt])
    return pp.copy(), pp.copy(), 1.0, copies


# ----------------------------------------------------------------------------
# the v3 outer loop (admm_solver_v3.py:621-733) on dict state
# ----------------------------------------------------------------------------
def admm_v3(case, max_it=1000, eps_abs=1e-4, eps_rel=1e-3, rho=1.0, verbose=False):
    from gcs_admm_amd.graph import chebyshev_center
    n = case["n"]; keys = [tuple(k) if isinstance(k, list) else k for k in case["keys"]]
    As = {k: np.array(a, float) for k, a in zip(keys, case["As"])}
    bs = {k: np.array(a, float) for k, a in zip(keys, case["bs"])}
    E = [(u, w) for u, w in case["edges"]]
    I_in = {v: [e for e in E if e[1] == v] for v in keys}
    I_out = {v: [e for e in E if e[0] == v] for v in keys}
    pint = {v: chebyshev_center(As[v], bs[v]) for v in keys}
    # coupled words per edge: zu (n), zw (n), y
    zed = {e: (np.zeros(n), np.zeros(n), 0.0) for e in E}
    cop = {(e, v): (np.zeros(n), np.zeros(n), 0.0) for e in E for v in e}
    mu = {(e, v): (np.zeros(n), np.zeros(n), 0.0) for e in E for v in e}
    nx = (4 * n + 1) * (len(keys) + 2 * len(E)); nmu = (4 * n + 2) * len(E)
    pri_seq = [0.0]; dual_seq = [0.0]; rho_seq = [rho]
    zv = {v: np.zeros(2 * n) for v in keys}; yv = {v: 0.0 for v in keys}; xv = {v: np.zeros(2 * n) for v in keys}
    it = 1; ipm_its = []
    while it <= max_it:
        for v in keys:
            inc_e = I_in[v] + I_out[v]
            inc = []
            for e in inc_e:
                m_ = mu[(e, v)]; z_ = zed[e]
                inc.append({"out": e[0] == v, "T_u": z_[0] - m_[0], "T_w": z_[1] - m_[1], "T_y": z_[2] - m_[2]})
            if v in ('s', 't'):
                x_, z_, y_, cps = solve_terminal(n, pint[v], inc, rho, v == 's')
            elif not I_in[v] or not I_out[v]:
                cps = [((np.zeros(n) if i["out"] else np.array(i["T_u"])), np.zeros(n), 0.0) for i in inc]
                x_, z_, y_ = np.concatenate([pint[v]] * 2), np.zeros(2 * n), 0.0
            else:
                x_, z_, y_, cps, its = solve_vertex_full(n, As[v], bs[v], pint[v], inc, rho)
                ipm_its.append(its)
            xv[v], zv[v], yv[v] = x_, z_, y_
            for e, cp in zip(inc_e, cps):
                cop[(e, v)] = cp
        zprev = dict(zed)
        s_r = s_dz = s_ax = s_bz = s_mu = 0.0
        for e in E:
            u, w = e
            cu, cw = cop[(e, u)], cop[(e, w)]
            znew = (0.5 * (cu[0] + cw[0]), 0.5 * (cu[1] + cw[1]), 0.5 * (cu[2] + cw[2]))
            zed[e] = znew
            for v, cv in ((u, cu), (w, cw)):
                r = (cv[0] - znew[0], cv[1] - znew[1], cv[2] - znew[2])
                m_ = mu[(e, v)]
                mu[(e, v)] = (m_[0] + r[0], m_[1] + r[1], m_[2] + r[2])
                s_r += r[0] @ r[0] + r[1] @ r[1] + r[2] ** 2
                s_ax += cv[0] @ cv[0] + cv[1] @ cv[1] + cv[2] ** 2
            dz = (znew[0] - zprev[e][0], znew[1] - zprev[e][1], znew[2] - zprev[e][2])
            s_dz += dz[0] @ dz[0] + dz[1] @ dz[1] + dz[2] ** 2
            s_bz += znew[0] @ znew[0] + znew[1] @ znew[1] + znew[2] ** 2
        pri = np.sqrt(s_r); dual = rho * np.sqrt(2 * s_dz)
        pri_seq.append(pri); dual_seq.append(dual)
        if pri >= 10 * dual and it < 100:
            rho *= 2; mu = {k: (m[0] / 2, m[1] / 2, m[2] / 2) for k, m in mu.items()}
        elif dual >= 10 * pri and it < 100:
            rho /= 2; mu = {k: (m[0] * 2, m[1] * 2, m[2] * 2) for k, m in mu.items()}
        rho_seq.append(rho)
        s_mu = sum(m[0] @ m[0] + m[1] @ m[1] + m[2] ** 2 for m in mu.values())
        e_pri = np.sqrt(nx) * eps_abs + eps_rel * max(np.sqrt(s_ax), np.sqrt(2 * s_bz))
        e_dual = np.sqrt(nmu) * eps_abs + eps_rel * np.sqrt(s_mu)
        if verbose and it % 10 == 0:
            print(it, pri, dual, e_pri, e_dual, flush=True)
        if pri < e_pri and dual < e_dual:
            break
        it += 1
    cost = sum(np.linalg.norm(z[:n] - z[n:]) for z in zv.values()) + 1e-4 * sum(z[2] for z in zed.values())
    return dict(iterations=it, pri=np.array(pri_seq), dual=np.array(dual_seq), rho=np.array(rho_seq),
                cost=cost, xv=xv, yv=yv, zv=zv, zed=zed, ipm_its=ipm_its)
