"""Oracle: hand-derived jet forward / adjoint of the per-instance residual MLP (TEST INFRASTRUCTURE).

The residual field of scripts/main.py:433-458 enters the renderer through its value AND its gradient w.r.t. the
sample position (the SDF normal, renderers.py:218-228), so the backward pass needs the adjoint of a first-order
"jet" (value + 3 tangents) pushed through

    fold (|x|, y, z)/100  ->  sinusoidal encoder (sinusoidal_encoder.py:12-18)  ->  Linear 49->16
    -> 3 x [LayerNorm(no affine) -> exact GELU -> Linear 17->16]  ->  LayerNorm -> GELU -> Linear 17->1
    -> sigmoid(. - 1)                                              (hyper_distance_field.py:57-73, main.py:446)

This is the float64 numpy blueprint of the HIP kernels (vsrd_amd/csrc/residual.h); tests/test_oracle_analytic.py checks
it against autograd through oracle.fields.residual_distance_and_gradient.
"""
import math

import numpy as np

WIDTHS = (48, 16, 16, 16, 16, 1)
SPLITS = tuple((WIDTHS[k] + 1) * WIDTHS[k + 1] for k in range(5))
SCALE = 100.0
LN_EPS = 1.0e-5
FREQS = (2.0 ** np.arange(8)) * math.pi


def _unpack(w):
    mats, off = [], 0
    for layer, count in enumerate(SPLITS):
        block = w[off:off + count].reshape(WIDTHS[layer + 1], WIDTHS[layer] + 1)
        mats.append((block[:, :-1], block[:, -1]))
        off += count
    return mats


def _phi(y):
    return np.exp(-0.5 * y * y) / math.sqrt(2.0 * math.pi)


def _Phi(y):
    from math import erf
    return 0.5 * (1.0 + np.vectorize(erf)(y / math.sqrt(2.0)))


def forward(p, w):
    """p [3], w [1617] -> (res, gres_p [3], cache)."""
    fold = np.array([np.sign(p[0]), 1.0, 1.0])
    f = np.array([abs(p[0]), p[1], p[2]]) / SCALE
    phase = f[:, None] * FREQS[None, :]                                  # [3,8]
    cos, sin = np.cos(phase), np.sin(phase)
    feat = np.stack([cos, sin], -1).reshape(48)                         # [coord][freq][cos,sin]
    dfeat = np.zeros((3, 48))
    for c in range(3):
        dfeat[c, c * 16:(c + 1) * 16] = np.stack([-FREQS * sin[c], FREQS * cos[c]], -1).reshape(16)
    mats = _unpack(w)
    W, b = mats[0]
    z, dz = W @ feat + b, dfeat @ W.T                                    # [16], [3,16]
    layers = []
    for W, b in mats[1:]:
        n = z.shape[0]
        mu = z.mean()
        s = math.sqrt(((z - mu) ** 2).mean() + LN_EPS)
        y = (z - mu) / s
        q = (dz * y).mean(-1)                                            # [3]
        dy = (dz - dz.mean(-1, keepdims=True) - y[None] * q[:, None]) / s
        g1 = _Phi(y) + y * _phi(y)
        a, da = y * _Phi(y), dy * g1[None]
        layers.append(dict(z=z, dz=dz, s=s, y=y, q=q, dy=dy, g1=g1, a=a, da=da, n=n))
        z, dz = W @ a + b, da @ W.T
    out, dout = z[0], dz[:, 0]
    res = 1.0 / (1.0 + math.exp(-(out - 1.0)))
    kappa = res * (1.0 - res)
    gres = kappa * dout * fold / SCALE
    return res, gres, dict(fold=fold, cos=cos, sin=sin, feat=feat, dfeat=dfeat, mats=mats, layers=layers, dout=dout, res=res, kappa=kappa)


def backward(p, w, res_bar, gres_bar):
    """Adjoint of forward(): (p_bar [3], w_bar [1617])."""
    res, _, c = forward(p, w)
    fold, kappa, dout, mats, layers = c["fold"], c["kappa"], c["dout"], c["mats"], c["layers"]
    dout_bar = gres_bar * kappa * fold / SCALE
    kappa_bar = float((gres_bar * dout * fold / SCALE).sum())
    out_bar = (res_bar + kappa_bar * (1.0 - 2.0 * res)) * kappa
    z_bar, dz_bar = np.array([out_bar]), dout_bar[:, None].copy()          # adjoint of the last linear's output jet
    w_bars = [None] * 5
    for layer in range(4, 0, -1):
        W, b = mats[layer]
        L = layers[layer - 1]
        # linear: z' = W a + b, dz'_c = W da_c
        W_bar = np.outer(z_bar, L["a"]) + np.einsum("co,ci->oi", dz_bar, L["da"])
        w_bars[layer] = np.concatenate([W_bar, z_bar[:, None]], axis=1).reshape(-1)
        a_bar, da_bar = W.T @ z_bar, dz_bar @ W
        # GELU
        y, dy, g1, s, n = L["y"], L["dy"], L["g1"], L["s"], L["n"]
        y_bar = a_bar * g1 + (da_bar * dy).sum(0) * _phi(y) * (2.0 - y * y)
        dy_bar = da_bar * g1[None]
        # LayerNorm jet
        P = lambda v: (v - v.mean(-1, keepdims=True) - y * (v * y).mean(-1, keepdims=True)) / s
        dz_prev_bar = P(dy_bar)
        y_bar = y_bar - (dy_bar * L["q"][:, None]).sum(0) / s - (L["dz"] * (dy_bar * y[None]).sum(-1, keepdims=True)).sum(0) / (n * s)
        s_bar = -(dy_bar * dy).sum() / s
        z_bar, dz_bar = P(y_bar) + s_bar * y / n, dz_prev_bar
    # first linear: z = W feat + b, dz_c = W dfeat_c
    W, b = mats[0]
    W_bar = np.outer(z_bar, c["feat"]) + np.einsum("co,ci->oi", dz_bar, c["dfeat"])
    w_bars[0] = np.concatenate([W_bar, z_bar[:, None]], axis=1).reshape(-1)
    feat_bar = W.T @ z_bar                                                   # [48]
    dfeat_bar = dz_bar @ W                                                   # [3,48] (only coordinate c's block of row c matters)
    f_bar = np.zeros(3)
    for cidx in range(3):
        fb = feat_bar[cidx * 16:(cidx + 1) * 16].reshape(8, 2)
        db = dfeat_bar[cidx, cidx * 16:(cidx + 1) * 16].reshape(8, 2)
        cos, sin = c["cos"][cidx], c["sin"][cidx]
        f_bar[cidx] = (fb[:, 0] * (-FREQS * sin) + fb[:, 1] * (FREQS * cos)).sum() \
                    + (db[:, 0] * (-FREQS ** 2 * cos) + db[:, 1] * (-FREQS ** 2 * sin)).sum()
    return f_bar * fold / SCALE, np.concatenate(w_bars)


def backward_directional(p, w, res_bar, gres_bar):
    """The same adjoint with ONE tangent instead of three (the form the HIP kernel uses, residual.h: residual_backward).

    The loss sees the three tangents of `out` only through the linear form  sum_c gres_bar_c kappa fold_c / 100 * dout_c, and a
    Jacobian-vector product is linear in its direction: with  delta = gres_bar * fold / 100  that form is  kappa * (tangent of out
    along delta).  So the backward pushes the single tangent  sum_c delta_c dfeat_c  through the network (seed kappa) instead of the
    three unit tangents (seeds kappa delta_c): half the hidden-layer work, same derivatives (delta is a constant of the adjoint:
    gres_bar is a seed)."""
    fold = np.array([np.sign(p[0]), 1.0, 1.0])
    f = np.array([abs(p[0]), p[1], p[2]]) / SCALE
    phase = f[:, None] * FREQS[None, :]
    cos, sin = np.cos(phase), np.sin(phase)
    feat = np.stack([cos, sin], -1).reshape(48)
    dfeat = np.stack([-FREQS[None, :] * sin, FREQS[None, :] * cos], -1).reshape(3, 16)            # d feat_c / d f_c, per coordinate block
    d2feat = -(np.repeat(FREQS, 2)[None, :] ** 2) * feat.reshape(3, 16)                           # d^2 feat_c / d f_c^2
    delta = gres_bar * fold / SCALE
    tangent = (delta[:, None] * dfeat).reshape(48)
    mats = _unpack(w)
    W, b = mats[0]
    z, dz = W @ feat + b, W @ tangent
    layers = []
    for W, b in mats[1:]:
        n = z.shape[0]
        mu = z.mean()
        s = math.sqrt(((z - mu) ** 2).mean() + LN_EPS)
        y = (z - mu) / s
        q = (dz * y).mean()
        dy = (dz - dz.mean() - y * q) / s
        g1 = _Phi(y) + y * _phi(y)
        layers.append(dict(dz=dz, s=s, y=y, q=q, dy=dy, g1=g1, a=y * _Phi(y), da=dy * g1, n=n))
        z, dz = W @ layers[-1]["a"] + b, W @ layers[-1]["da"]
    out, dout = z[0], dz[0]                                                                       # dout = kappa_bar of backward()
    res = 1.0 / (1.0 + math.exp(-(out - 1.0)))
    kappa = res * (1.0 - res)
    z_bar, dz_bar = np.array([(res_bar + dout * (1.0 - 2.0 * res)) * kappa]), np.array([kappa])
    w_bars = [None] * 5
    for layer in range(4, 0, -1):
        W, b = mats[layer]
        L = layers[layer - 1]
        W_bar = np.outer(z_bar, L["a"]) + np.outer(dz_bar, L["da"])
        w_bars[layer] = np.concatenate([W_bar, z_bar[:, None]], axis=1).reshape(-1)
        a_bar, da_bar = W.T @ z_bar, W.T @ dz_bar
        y, dy, g1, s, n = L["y"], L["dy"], L["g1"], L["s"], L["n"]
        P = lambda v: (v - v.mean() - y * (v * y).mean()) / s
        dy_bar = da_bar * g1
        y_bar = a_bar * g1 + da_bar * dy * _phi(y) * (2.0 - y * y) - dy_bar * L["q"] / s - L["dz"] * (dy_bar * y).sum() / (n * s)
        s_bar = -(dy_bar * dy).sum() / s
        z_bar, dz_bar = P(y_bar) + s_bar * y / n, P(dy_bar)
    W, b = mats[0]
    W_bar = np.outer(z_bar, feat) + np.outer(dz_bar, tangent)
    w_bars[0] = np.concatenate([W_bar, z_bar[:, None]], axis=1).reshape(-1)
    feat_bar, tangent_bar = (W.T @ z_bar).reshape(3, 16), (W.T @ dz_bar).reshape(3, 16)
    f_bar = (feat_bar * dfeat).sum(-1) + delta * (tangent_bar * d2feat).sum(-1)
    return f_bar * fold / SCALE, np.concatenate(w_bars)


def forward_reverse(p, w):
    """Value and gradient of the residual by ONE forward column and ONE reverse column (instead of value + three forward tangents):
    z_l -> y_l = LN(z_l) -> a_l = gelu(y_l) -> z_{l+1}; then  a_bar_3 = w4,  z_bar_l = P_l(a_bar_l * gelu'(y_l)),
    a_bar_{l-1} = W_l^T z_bar_l,  feat_bar = W_0^T z_bar_0,  d out / d f_c = sum_j feat_bar_{c,j} dfeat_{c,j}.
    Returns (res, gres_p [3]) like forward()."""
    fold = np.array([np.sign(p[0]), 1.0, 1.0])
    f = np.array([abs(p[0]), p[1], p[2]]) / SCALE
    phase = f[:, None] * FREQS[None, :]
    cos, sin = np.cos(phase), np.sin(phase)
    feat = np.stack([cos, sin], -1).reshape(48)
    dfeat = np.stack([-FREQS[None, :] * sin, FREQS[None, :] * cos], -1).reshape(3, 16)
    mats = _unpack(w)
    W, b = mats[0]
    z = W @ feat + b
    states = []
    for W, b in mats[1:]:
        mu = z.mean()
        s = math.sqrt(((z - mu) ** 2).mean() + LN_EPS)
        y = (z - mu) / s
        states.append((y, s, _Phi(y) + y * _phi(y)))
        z = W @ (y * _Phi(y)) + b
    out = z[0]
    a_bar = mats[4][0][0].copy()                                                                  # w4
    for layer in range(3, -1, -1):
        y, s, g1 = states[layer]
        v = a_bar * g1
        z_bar = (v - v.mean() - y * (v * y).mean()) / s
        a_bar = mats[layer][0].T @ z_bar
    dout = (a_bar.reshape(3, 16) * dfeat).sum(-1)
    res = 1.0 / (1.0 + math.exp(-(out - 1.0)))
    return res, res * (1.0 - res) * dout * fold / SCALE
