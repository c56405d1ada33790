"""Oracle: hand-derived adjoint of the box-only renderer (TEST INFRASTRUCTURE).

The reference obtains its parameter gradients by autograd, including a double-backward
through ``torch.autograd.grad(sdf, positions, create_graph=True)`` (renderers.py:218-228).
The HIP backward kernel cannot use autograd, so DESIGN.md §"Backward" derives the adjoint
in closed form.  This module is that derivation written in numpy float64, one ray at a
time; ``tests/test_oracle_analytic.py`` checks it against ordinary autograd through
``oracle.rendering`` so the formulas are verified on the CPU before they are transcribed
into ``vsrd_amd/csrc/render_backward.hip``.

Two phases, exactly as in the kernel:
  phase A: per sample, one sweep over the instances accumulating the soft-min sums, then the
           per-sample adjoint chain  labels -> weights -> opacity -> (u, cos) -> (u_bar, g_bar);
  phase B: per (sample, instance), adjoint of (d_i, grad d_i) w.r.t. (t_i, R_i, dim_i).
"""
import numpy as np

NORM_EPSILON = 1.0e-6


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def instance_geometry(x, t, R, dim):
    """x [S,3]; t [N,3]; R [N,3,3]; dim [N,3] -> dict of [S,N,...] arrays (sdfs.py:9-37)."""
    rel = x[:, None, :] - t[None, :, :]
    p = np.einsum("snk,nkj->snj", rel, R)
    q = np.abs(p) - dim[None]
    a = np.maximum(q, 0.0)
    nrm = np.sqrt((a * a).sum(-1) + NORM_EPSILON)
    arg = q.argmax(-1)
    m = np.take_along_axis(q, arg[..., None], -1)[..., 0]
    d = nrm - np.maximum(-m, 0.0)
    onehot = (np.arange(3)[None, None, :] == arg[..., None]) * (m < 0)[..., None]
    h = a / nrm[..., None] + onehot
    gl = np.sign(p) * h
    gw = np.einsum("snj,nkj->snk", gl, R)
    return dict(rel=rel, p=p, q=q, a=a, nrm=nrm, d=d, h=h, gl=gl, gw=gw)


def forward_ray(o, r, dist, t, R, dim, temperature, std, ratio, eps=1.0e-6):
    delta = dist[1:] - dist[:-1]
    mid = (dist[:-1] + dist[1:]) / 2.0
    x = o[None, :] + r[None, :] * mid[:, None]
    geo = instance_geometry(x, t, R, dim)
    d, gw = geo["d"], geo["gw"]
    dmin = d.min(-1, keepdims=True)
    e = np.exp(-(d - dmin) / temperature)
    Z = e.sum(-1, keepdims=True)
    w = e / Z
    u = (w * d).sum(-1)
    c = w * (1.0 - (d - u[:, None]) / temperature)
    g = (c[..., None] * gw).sum(1)
    gn = np.maximum(np.linalg.norm(g, axis=-1), 1.0e-12)
    n = g / gn[:, None]
    cos = n @ r
    A1, B1 = np.maximum(0.5 - 0.5 * cos, 0.0), np.maximum(-cos, 0.0)
    cprime = -(A1 + ratio * (B1 - A1))
    hh = cprime * delta / 2.0
    phi_p, phi_n = _sigmoid((u - hh) / std), _sigmoid((u + hh) / std)
    xx = (phi_p - phi_n) / (phi_p + eps)
    alpha = np.maximum(xx, 0.0)
    trans = np.concatenate([[1.0], np.cumprod(1.0 - alpha)[:-1]])
    wgt = trans * alpha
    labels = (wgt[:, None] * w).sum(0)
    return dict(geo=geo, x=x, delta=delta, w=w, u=u, c=c, g=g, gn=gn, n=n, cos=cos, phi_p=phi_p, phi_n=phi_n,
                xx=xx, alpha=alpha, trans=trans, wgt=wgt, labels=labels)


def backward_ray(o, r, dist, t, R, dim, temperature, std, ratio, lam, gamma=None, omega=None, eps=1.0e-6):
    """Adjoint of (labels, gradients, weights) w.r.t. (t, R, dim) for one ray.

    lam [N] = dL/dlabels, gamma [S',3] = dL/dgradients (or None), omega [S'] = dL/dweights (or None).
    """
    f = forward_ray(o, r, dist, t, R, dim, temperature, std, ratio, eps)
    geo, w, c, g, n = f["geo"], f["w"], f["c"], f["g"], f["n"]
    wgt, trans, alpha = f["wgt"], f["trans"], f["alpha"]
    T = temperature
    # ---- phase A: per-sample adjoints -------------------------------------------------------
    Lam = w @ lam                                            # sum_n lam_n w_{s,n}
    w_bar = Lam + (0.0 if omega is None else omega)
    contrib = w_bar * wgt
    Q = np.concatenate([np.cumsum(contrib[::-1])[::-1][1:], [0.0]])   # sum_{k>s}
    alpha_bar = w_bar * trans - Q / (1.0 - alpha)
    x_bar = alpha_bar * (f["xx"] > 0)
    phi_p, phi_n = f["phi_p"], f["phi_n"]
    phi_p_bar = x_bar * (phi_n + eps) / (phi_p + eps) ** 2
    phi_n_bar = -x_bar / (phi_p + eps)
    sp_bar = phi_p_bar * phi_p * (1.0 - phi_p) / std
    sn_bar = phi_n_bar * phi_n * (1.0 - phi_n) / std
    u_bar = sp_bar + sn_bar
    cprime_bar = (sn_bar - sp_bar) * f["delta"] / 2.0
    dcp_dcos = (1.0 - ratio) * 0.5 * (0.5 - 0.5 * f["cos"] > 0) + ratio * (-f["cos"] > 0)
    n_bar = (cprime_bar * dcp_dcos)[:, None] * r[None, :]
    g_bar = (n_bar - n * (n * n_bar).sum(-1, keepdims=True)) / f["gn"][:, None]
    if gamma is not None:
        g_bar = g_bar + gamma
    A = (g_bar * g).sum(-1)
    gbar0 = (w[..., None] * geo["gw"]).sum(1)
    B = (g_bar * gbar0).sum(-1)
    # ---- phase B: per (sample, instance) ----------------------------------------------------
    beta = np.einsum("sk,snk->sn", g_bar, geo["gw"])
    w_hat = lam[None, :] * wgt[:, None]
    W_bar = (wgt * Lam)[:, None]
    d_bar = (u_bar[:, None] * c
             + (-beta * c + w * A[:, None] - beta * w + c * B[:, None]) / T
             - w * (w_hat - W_bar) / T)
    gw_bar = c[..., None] * g_bar[:, None, :]
    gl_bar = np.einsum("nkj,snk->snj", R, gw_bar)
    v = np.sign(geo["p"]) * gl_bar
    hhat = geo["a"] / geo["nrm"][..., None]
    q_bar = d_bar[..., None] * geo["h"] + (geo["q"] > 0) * (v - hhat * (hhat * v).sum(-1, keepdims=True)) / geo["nrm"][..., None]
    p_bar = np.sign(geo["p"]) * q_bar
    grad_dim = -q_bar.sum(0)
    grad_R = np.einsum("snk,snj->nkj", geo["rel"], p_bar) + np.einsum("snk,snj->nkj", gw_bar, geo["gl"])
    grad_t = -np.einsum("nkj,snj->nk", R, p_bar)
    return grad_t, grad_R, grad_dim, f
