"""TEST INFRASTRUCTURE ONLY -- numpy closed-form restatement of the MRLA hot path.

Forward *and* hand-derived backward of every row of SURVEY.md section 8(a), in plain numpy so that
it can be evaluated in fp64 (tight reference) or fp32.  It is independent of torch autograd: the
backward formulas here are the ones the HIP kernels implement, and ``tests/`` check them against
gradients recorded from the reference's autograd (``tests/golden/*.npz``, produced by
``oracle/make_goldens.py``).

Reference lines restated (paths relative to /root/reference):
  a1  resnet/models/modules/mrla_light_module.py:52-74   light layer
  a2  resnet/models/resnet_mrla_light.py:40-43           + lambda_t * o_{t-1}
  a3  resnet/models/resnet_mrla_light.py:113-116         x + DropPath(BN(m))
  a4  resnet/models/modules/mrla_base_module.py:54-89    base layer (softmax over depth)
  a5  resnet/models/resnet_mrla_base.py:120-129          x + DropPath(relu(BN(attn)))
  a6  deit/deit_mrla_light.py:157-180                    light layer with GELU on V
  a7  deit/deit_mrla_light.py:194-209, 227-235           token module (2x LayerNorm, cls passthrough)
  --  resnet/models/utils/drop.py:17-24                  per-sample drop-path scale

Layout everywhere: x[b, c, h, w]; tokens [b, n, c].
"""
from math import log, sqrt, erf, pi

import numpy as np


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def k_size_for(c):
    """Conv1d tap count used for Wq/Wk (mrla_light_module.py:40-42)."""
    t = int(abs((log(c, 2) + 1) / 2.0))
    return t if t % 2 else t + 1


def corr1d(y, w):
    """out[b, c] = sum_j w[j] * y[b, c + j - p], zero padded (nn.Conv1d(1,1,k,padding=p))."""
    k = w.shape[0]
    p = (k - 1) // 2
    b, c = y.shape
    yp = np.zeros((b, c + 2 * p), dtype=y.dtype)
    yp[:, p:p + c] = y
    out = np.zeros_like(y)
    for j in range(k):
        out += w[j] * yp[:, j:j + c]
    return out


def corr1d_T(d, w):
    """Adjoint of corr1d wrt y: dy[b, c] = sum_j w[j] * d[b, c - j + p]."""
    k = w.shape[0]
    p = (k - 1) // 2
    b, c = d.shape
    dp = np.zeros((b, c + 2 * p), dtype=d.dtype)
    dp[:, p:p + c] = d
    out = np.zeros_like(d)
    for j in range(k):
        # index c - j + p  ->  padded index c - j + 2p
        out += w[j] * dp[:, 2 * p - j:2 * p - j + c]
    return out


def corr1d_wgrad(d, y, k):
    """dw[j] = sum_{b,c} d[b,c] * y[b, c + j - p]."""
    p = (k - 1) // 2
    b, c = y.shape
    yp = np.zeros((b, c + 2 * p), dtype=y.dtype)
    yp[:, p:p + c] = y
    return np.array([(d * yp[:, j:j + c]).sum() for j in range(k)], dtype=y.dtype)


def _pad_hw(x):
    b, c, h, w = x.shape
    xp = np.zeros((b, c, h + 2, w + 2), dtype=x.dtype)
    xp[:, :, 1:h + 1, 1:w + 1] = x
    return xp


def dwconv3x3(x, wv):
    """V[b,c,h,w] = sum_{i,j} wv[c,i,j] * x[b,c,h+i-1,w+j-1] (groups=c, pad 1, cross-correlation)."""
    b, c, h, w = x.shape
    xp = _pad_hw(x)
    out = np.zeros_like(x)
    for i in range(3):
        for j in range(3):
            out += wv[None, :, i, j, None, None] * xp[:, :, i:i + h, j:j + w]
    return out


def dwconv3x3_T(dv, wv):
    """dx[b,c,h,w] = sum_{i,j} wv[c,i,j] * dv[b,c,h-i+1,w-j+1]."""
    b, c, h, w = dv.shape
    dp = _pad_hw(dv)
    out = np.zeros_like(dv)
    for i in range(3):
        for j in range(3):
            out += wv[None, :, i, j, None, None] * dp[:, :, 2 - i:2 - i + h, 2 - j:2 - j + w]
    return out


def dwconv3x3_wgrad(dv, x):
    """dwv[c,i,j] = sum_{b,h,w} dv[b,c,h,w] * x[b,c,h+i-1,w+j-1]."""
    b, c, h, w = x.shape
    xp = _pad_hw(x)
    out = np.zeros((c, 3, 3), dtype=x.dtype)
    for i in range(3):
        for j in range(3):
            out[:, i, j] = (dv * xp[:, :, i:i + h, j:j + w]).sum(axis=(0, 2, 3))
    return out


_verf = np.vectorize(erf, otypes=[np.float64])


def gelu(v):
    """Exact (erf) GELU, as nn.GELU() default (deit_mrla_light.py:153)."""
    return (0.5 * v * (1.0 + _verf(v / sqrt(2.0)))).astype(v.dtype)


def gelu_grad(v):
    cdf = 0.5 * (1.0 + _verf(v / sqrt(2.0)))
    pdf = np.exp(-0.5 * v.astype(np.float64) ** 2) / sqrt(2.0 * pi)
    return (cdf + v * pdf).astype(v.dtype)


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def head_expand(a, d):
    """[b, g] -> [b, g*d] (heads are contiguous channel groups)."""
    return np.repeat(a, d, axis=1)


def drop_path_scale(keep_mask, p):
    """Per-sample multiplier of drop.py:17-24: floor(keep_prob + U) / keep_prob, given the 0/1 mask."""
    return keep_mask.astype(np.float64) / (1.0 - p)


# ----------------------------------------------------------------------------------------------
# a1 / a6: light layer
# ----------------------------------------------------------------------------------------------
def light_layer_fwd(x, wq, wk, wv, d, act_gelu=False):
    """Returns (out, cache).  wq, wk: [k]; wv: [c,3,3]; d = channels per head."""
    b, c, h, w = x.shape
    g = c // d
    s = 1.0 / sqrt(c / g)
    y = x.mean(axis=(2, 3))
    q = corr1d(y, wq)
    kk = corr1d(y, wk)
    logit = (q * kk).reshape(b, g, d).sum(-1) * s
    a = sigmoid(logit)
    u = dwconv3x3(x, wv)
    v = gelu(u) if act_gelu else u
    out = head_expand(a, d)[:, :, None, None] * v
    cache = dict(x=x, wq=wq, wk=wk, wv=wv, d=d, s=s, y=y, q=q, kk=kk, a=a, u=u, v=v,
                 act_gelu=act_gelu)
    return out, cache


def light_layer_bwd(dout, cache):
    """Returns dict(dx, dwq, dwk, dwv) for upstream gradient dout."""
    x, wq, wk, wv, d, s = (cache[n] for n in ("x", "wq", "wk", "wv", "d", "s"))
    y, q, kk, a, u, v = (cache[n] for n in ("y", "q", "kk", "a", "u", "v"))
    b, c, h, w = x.shape
    g = c // d
    ae = head_expand(a, d)[:, :, None, None]
    dv = ae * dout
    da = (dout * v).sum(axis=(2, 3)).reshape(b, g, d).sum(-1)
    du = dv * gelu_grad(u) if cache["act_gelu"] else dv
    dlogit = da * a * (1.0 - a) * s
    dle = head_expand(dlogit, d)
    dq = dle * kk
    dkk = dle * q
    dwq = corr1d_wgrad(dq, y, wq.shape[0])
    dwk = corr1d_wgrad(dkk, y, wk.shape[0])
    dy = corr1d_T(dq, wq) + corr1d_T(dkk, wk)
    dx = dwconv3x3_T(du, wv) + dy[:, :, None, None] / (h * w)
    dwv = dwconv3x3_wgrad(du, x)
    return dict(dx=dx, dwq=dwq, dwk=dwk, dwv=dwv)


# ----------------------------------------------------------------------------------------------
# BatchNorm2d (train: batch stats, biased var for normalisation, unbiased for running update)
# ----------------------------------------------------------------------------------------------
def bn_fwd(m, gamma, beta, run_mean, run_var, training, eps=1e-5, momentum=0.1):
    b, c, h, w = m.shape
    if training:
        mu = m.mean(axis=(0, 2, 3))
        var = m.var(axis=(0, 2, 3))
        n = b * h * w
        new_rm = (1 - momentum) * run_mean + momentum * mu
        new_rv = (1 - momentum) * run_var + momentum * var * n / max(n - 1, 1)
    else:
        mu, var = run_mean, run_var
        new_rm, new_rv = run_mean, run_var
    inv = 1.0 / np.sqrt(var + eps)
    mhat = (m - mu[None, :, None, None]) * inv[None, :, None, None]
    z = gamma[None, :, None, None] * mhat + beta[None, :, None, None]
    return z, dict(mhat=mhat, inv=inv, gamma=gamma, training=training, new_rm=new_rm, new_rv=new_rv)


def bn_bwd(dz, cache):
    mhat, inv, gamma = cache["mhat"], cache["inv"], cache["gamma"]
    dgamma = (dz * mhat).sum(axis=(0, 2, 3))
    dbeta = dz.sum(axis=(0, 2, 3))
    gi = (gamma * inv)[None, :, None, None]
    if cache["training"]:
        n = dz.shape[0] * dz.shape[2] * dz.shape[3]
        dm = gi * (dz - (dbeta / n)[None, :, None, None] - mhat * (dgamma / n)[None, :, None, None])
    else:
        dm = gi * dz
    return dm, dgamma, dbeta


# ----------------------------------------------------------------------------------------------
# a2 + a3: light block tail   out = x + dp[b] * BN(light(x) + lambda * o_prev)
# ----------------------------------------------------------------------------------------------
def light_tail_fwd(x, o_prev, wq, wk, wv, lam, gamma, beta, run_mean, run_var, d,
                   training=True, dp=None, eps=1e-5, momentum=0.1):
    """dp: per-sample drop-path multiplier [b] (None = identity).  Returns (out, cache)."""
    attn, c1 = light_layer_fwd(x, wq, wk, wv, d)
    m = attn + lam[None, :, None, None] * o_prev
    z, c2 = bn_fwd(m, gamma, beta, run_mean, run_var, training, eps, momentum)
    dpv = np.ones(x.shape[0], dtype=x.dtype) if dp is None else dp.astype(x.dtype)
    out = x + dpv[:, None, None, None] * z
    return out, dict(layer=c1, bn=c2, lam=lam, o_prev=o_prev, dp=dpv, m=m)


def light_tail_bwd(dout, cache):
    """Returns dict(dx, do_prev, dwq, dwk, dwv, dlam, dgamma, dbeta)."""
    dz = cache["dp"][:, None, None, None] * dout
    dm, dgamma, dbeta = bn_bwd(dz, cache["bn"])
    gl = light_layer_bwd(dm, cache["layer"])
    lam = cache["lam"]
    return dict(dx=dout + gl["dx"], do_prev=lam[None, :, None, None] * dm,
                dwq=gl["dwq"], dwk=gl["dwk"], dwv=gl["dwv"],
                dlam=(dm * cache["o_prev"]).sum(axis=(0, 2, 3)), dgamma=dgamma, dbeta=dbeta)


# ----------------------------------------------------------------------------------------------
# a4: base layer over a growing K/V history
# ----------------------------------------------------------------------------------------------
def _ident(a):
    return a


def base_layer_fwd(x, wq, wk, wv, d, K_prev=None, V_prev=None, rnd=_ident, rnd_dv=_ident):
    """K_prev [b,t-1,c] / V_prev [b,t-1,c,h,w] or None (init_cell).  Returns (out, K, V, cache).
    `rnd` models the storage dtype of the HIP path (identity for fp32; round-to-bf16 for bf16 activations):
    it is applied where that path stores an activation-sized tensor (v_t, attn)."""
    b, c, h, w = x.shape
    g = c // d
    s = 1.0 / sqrt(c / g)
    y = x.mean(axis=(2, 3))
    q = corr1d(y, wq)
    kt = corr1d(y, wk)
    vt = rnd(dwconv3x3(x, wv))
    if K_prev is None:
        K = kt[:, None]
        V = vt[:, None]
    else:
        K = np.concatenate([K_prev, kt[:, None]], axis=1)
        V = np.concatenate([V_prev, vt[:, None]], axis=1)
    t = K.shape[1]
    logits = np.einsum("bgd,btgd->bgt", q.reshape(b, g, d), K.reshape(b, t, g, d)) * s
    logits = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(logits)
    P = e / e.sum(axis=-1, keepdims=True)                       # [b,g,t]
    out = rnd(np.einsum("bgt,btgdhw->bgdhw", P, V.reshape(b, t, g, d, h, w)).reshape(b, c, h, w))
    cache = dict(x=x, wq=wq, wk=wk, wv=wv, d=d, s=s, y=y, q=q, K=K, V=V, P=P, rnd=rnd, rnd_dv=rnd_dv)
    return out, K, V, cache


def base_layer_bwd(dout, dK, dV, cache):
    """dout: grad of `out`; dK [b,t,c] / dV [b,t,c,h,w]: grads flowing into the returned K, V from
    later layers (zeros for the last layer).  Returns dict(dx, dwq, dwk, dwv, dK_prev, dV_prev)."""
    x, wq, wk, wv, d, s = (cache[n] for n in ("x", "wq", "wk", "wv", "d", "s"))
    y, q, K, V, P = (cache[n] for n in ("y", "q", "K", "V", "P"))
    b, c, h, w = x.shape
    g = c // d
    t = K.shape[1]
    dout = cache["rnd"](dout)                                    # dA_t as stored in the dA ring
    do = dout.reshape(b, g, d, h, w)
    dP = np.einsum("bgdhw,btgdhw->bgt", do, V.reshape(b, t, g, d, h, w))
    dVtot = dV + np.einsum("bgt,bgdhw->btgdhw", P, do).reshape(b, t, c, h, w)
    dlog = P * (dP - (P * dP).sum(-1, keepdims=True)) * s        # [b,g,t]
    dq = np.einsum("bgt,btgd->bgd", dlog, K.reshape(b, t, g, d)).reshape(b, c)
    dKtot = dK + np.einsum("bgt,bgd->btgd", dlog, q.reshape(b, g, d)).reshape(b, t, c)
    dkt = dKtot[:, -1]
    dvt = cache.get("rnd_dv", _ident)(dVtot[:, -1])            # dV_t as the channels_last path stores it between its two kernels
    dwq = corr1d_wgrad(dq, y, wq.shape[0])
    dwk = corr1d_wgrad(dkt, y, wk.shape[0])
    dy = corr1d_T(dq, wq) + corr1d_T(dkt, wk)
    dx = dwconv3x3_T(dvt, wv) + dy[:, :, None, None] / (h * w)
    dwv = dwconv3x3_wgrad(dvt, x)
    return dict(dx=dx, dwq=dwq, dwk=dwk, dwv=dwv,
                dK_prev=dKtot[:, :-1] if t > 1 else None, dV_prev=dVtot[:, :-1] if t > 1 else None)


# ----------------------------------------------------------------------------------------------
# a5: base block tail   out = x + dp[b] * relu(BN(attn))
# ----------------------------------------------------------------------------------------------
def base_tail_fwd(x, wq, wk, wv, gamma, beta, run_mean, run_var, d, K_prev=None, V_prev=None,
                  training=True, dp=None, eps=1e-5, momentum=0.1, rnd=_ident, rnd_dv=_ident):
    attn, K, V, c1 = base_layer_fwd(x, wq, wk, wv, d, K_prev, V_prev, rnd, rnd_dv)
    z, c2 = bn_fwd(attn, gamma, beta, run_mean, run_var, training, eps, momentum)
    r = np.maximum(z, 0)
    dpv = np.ones(x.shape[0], dtype=x.dtype) if dp is None else dp.astype(x.dtype)
    out = x + dpv[:, None, None, None] * r
    return out, K, V, dict(layer=c1, bn=c2, z=z, dp=dpv)


def base_tail_bwd(dout, dK, dV, cache):
    dr = cache["dp"][:, None, None, None] * dout
    dz = dr * (cache["z"] > 0)
    dattn, dgamma, dbeta = bn_bwd(dz, cache["bn"])
    gl = base_layer_bwd(dattn, dK, dV, cache["layer"])
    gl["dx"] = gl["dx"] + dout
    gl["dgamma"], gl["dbeta"] = dgamma, dbeta
    return gl


# ----------------------------------------------------------------------------------------------
# LayerNorm over the last axis
# ----------------------------------------------------------------------------------------------
def ln_fwd(x, w, bias, eps=1e-6):
    mu = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    inv = 1.0 / np.sqrt(var + eps)
    xh = (x - mu) * inv
    return xh * w + bias, dict(xh=xh, inv=inv, w=w)


def ln_bwd(dy, cache):
    xh, inv, w = cache["xh"], cache["inv"], cache["w"]
    red = tuple(range(dy.ndim - 1))
    dw = (dy * xh).sum(axis=red)
    db = dy.sum(axis=red)
    dxh = dy * w
    dx = inv * (dxh - dxh.mean(-1, keepdims=True) - xh * (dxh * xh).mean(-1, keepdims=True))
    return dx, dw, db


# ----------------------------------------------------------------------------------------------
# a7: DeiT token module   out = cat(cls(LN_x x), light_gelu(tokens(LN_x x)) + lambda * LN_o(o)[1:])
# ----------------------------------------------------------------------------------------------
def token_light_fwd(xt, ot, lnx_w, lnx_b, lno_w, lno_b, wq, wk, wv, lam, d, eps=1e-6):
    """xt, ot: [b, n, c] with n-1 a perfect square.  Returns (out[b,n,c], cache)."""
    b, n, c = xt.shape
    side = int(sqrt(n - 1))
    xn, cx = ln_fwd(xt, lnx_w, lnx_b, eps)
    on, co = ln_fwd(ot, lno_w, lno_b, eps)
    fmap = xn[:, 1:].reshape(b, side, side, c).transpose(0, 3, 1, 2)
    attn, cl = light_layer_fwd(np.ascontiguousarray(fmap), wq, wk, wv, d, act_gelu=True)
    tok = attn.reshape(b, c, side * side).transpose(0, 2, 1) + lam * on[:, 1:]
    out = np.concatenate([xn[:, :1], tok], axis=1)
    return out, dict(cx=cx, co=co, cl=cl, lam=lam, on=on, side=side)


def token_light_bwd(dout, cache):
    """Returns dict(dxt, dot, dlnx_w, dlnx_b, dlno_w, dlno_b, dwq, dwk, dwv, dlam)."""
    b, n, c = dout.shape
    side, lam, on = cache["side"], cache["lam"], cache["on"]
    dtok = dout[:, 1:]
    dattn = dtok.transpose(0, 2, 1).reshape(b, c, side, side)
    gl = light_layer_bwd(np.ascontiguousarray(dattn), cache["cl"])
    dxn = np.empty_like(dout)
    dxn[:, :1] = dout[:, :1]
    dxn[:, 1:] = gl["dx"].reshape(b, c, side * side).transpose(0, 2, 1)
    don = np.zeros_like(dout)
    don[:, 1:] = lam * dtok
    dxt, dlnx_w, dlnx_b = ln_bwd(dxn, cache["cx"])
    dot, dlno_w, dlno_b = ln_bwd(don, cache["co"])
    return dict(dxt=dxt, dot=dot, dlnx_w=dlnx_w, dlnx_b=dlnx_b, dlno_w=dlno_w, dlno_b=dlno_b,
                dwq=gl["dwq"], dwk=gl["dwk"], dwv=gl["dwv"],
                dlam=(dtok * on[:, 1:]).sum(axis=(0, 1)))
