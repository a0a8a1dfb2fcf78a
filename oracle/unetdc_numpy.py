"""CPU oracle for the U-Net / U-Net-DC forward+backward path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the arithmetic the reference delegates to PyTorch
ATen.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package (``unet_dc_segmentation_amd``) never
imports anything under ``oracle/`` and fails loudly when its HIP library is missing.

Parity pinning: every function here is checked against the live reference
(``/root/reference/models/model_2.py`` + ``utils/metrics_DC.py`` imported in the build
container) by ``tools/make_goldens.py``; the resulting input/output vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py`` (no reference needed at test
time).  The reference itself ships no tests or golden vectors for this path,
so those generated fixtures are the pin.

All tensors are NCHW numpy arrays; ``dtype`` follows the inputs (float32 mirrors the reference,
float64 gives a high-precision yardstick).  Citations are to files under /root/reference.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5        # nn.BatchNorm2d default used at models/model_2.py:45,52
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default

BLOCKS = ("enc1", "enc2", "enc3", "enc4", "bottleneck", "dec4", "dec3", "dec2", "dec1")
DILATIONS_DC = {"enc1": 1, "enc2": 2, "enc3": 4, "enc4": 8, "bottleneck": 16,   # model_2.py:10-16
                "dec4": 1, "dec3": 1, "dec2": 1, "dec1": 1}                      # model_2.py:21-30
DILATIONS_PLAIN = {b: 1 for b in BLOCKS}                                         # models/model.py:25-33


# --------------------------------------------------------------------------- conv 3x3 (dilated)
def _shifted(x, dy, dx):
    """x[n,c,y+dy,x+dx] with zero fill outside the image (zero padding = dilation, model_2.py:41-44)."""
    n, c, h, w = x.shape
    out = np.zeros_like(x)
    ys0, ys1 = max(0, -dy), min(h, h - dy)
    xs0, xs1 = max(0, -dx), min(w, w - dx)
    if ys0 < ys1 and xs0 < xs1:
        out[:, :, ys0:ys1, xs0:xs1] = x[:, :, ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx]
    return out


def conv3x3(x, w, b, d):
    """nn.Conv2d(k=3, padding=d, dilation=d) forward (model_2.py:41-44,48-51).

    y[n,co,y,x] = b[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * x[n,ci,y+(ky-1)d,x+(kx-1)d]."""
    n, ci, h, wd = x.shape
    co = w.shape[0]
    y = np.zeros((n, co, h, wd), dtype=x.dtype)
    for ky in range(3):
        for kx in range(3):
            xs = _shifted(x, (ky - 1) * d, (kx - 1) * d)
            y += np.einsum("oc,nchw->nohw", w[:, :, ky, kx], xs, optimize=True)
    if b is not None:
        y += b.reshape(1, co, 1, 1)
    return y


def conv3x3_bwd(x, w, d, gy, need_gx=True):
    """Autograd of conv3x3: returns (gx, gw, gb)."""
    gw = np.zeros_like(w)
    gx = np.zeros_like(x) if need_gx else None
    for ky in range(3):
        for kx in range(3):
            dy, dx = (ky - 1) * d, (kx - 1) * d
            xs = _shifted(x, dy, dx)
            gw[:, :, ky, kx] = np.einsum("nohw,nchw->oc", gy, xs, optimize=True)
            if need_gx:
                # gx[p] += sum_co gy[p - off] * w  -> shift gy by -off
                gys = _shifted(gy, -dy, -dx)
                gx += np.einsum("oc,nohw->nchw", w[:, :, ky, kx], gys, optimize=True)
    gb = gy.sum(axis=(0, 2, 3))
    return gx, gw, gb


# --------------------------------------------------------------------------- batch norm + relu
def bn_train(x, gamma, beta, eps=BN_EPS):
    """nn.BatchNorm2d in train mode (model_2.py:45,52): biased batch variance over (N,H,W).

    Returns y and the cache (xhat, rstd, mean, var_biased)."""
    mean = x.mean(axis=(0, 2, 3))
    var = x.var(axis=(0, 2, 3))                     # biased (ddof=0)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean.reshape(1, -1, 1, 1)) * rstd.reshape(1, -1, 1, 1)
    y = xhat * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    return y.astype(x.dtype), (xhat.astype(x.dtype), rstd, mean, var)


def bn_running_update(running_mean, running_var, mean, var_biased, count, momentum=BN_MOMENTUM):
    """running stats update: running_var uses the UNBIASED batch variance."""
    unbiased = var_biased * (count / max(count - 1, 1))
    return ((1 - momentum) * running_mean + momentum * mean,
            (1 - momentum) * running_var + momentum * unbiased)


def bn_train_bwd(gy, cache, gamma):
    xhat, rstd, _, _ = cache
    m = gy.shape[0] * gy.shape[2] * gy.shape[3]
    gbeta = gy.sum(axis=(0, 2, 3))
    ggamma = (gy * xhat).sum(axis=(0, 2, 3))
    k = (gamma * rstd).reshape(1, -1, 1, 1)
    gx = k * (gy - gbeta.reshape(1, -1, 1, 1) / m - xhat * ggamma.reshape(1, -1, 1, 1) / m)
    return gx.astype(gy.dtype), ggamma, gbeta


def bn_eval(x, gamma, beta, rm, rv, eps=BN_EPS):
    """nn.BatchNorm2d in eval mode: running statistics."""
    s = gamma / np.sqrt(rv + eps)
    return (x - rm.reshape(1, -1, 1, 1)) * s.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)


def relu(x):
    """nn.ReLU (model_2.py:46,53)."""
    return np.maximum(x, 0)


# --------------------------------------------------------------------------- max pool 2x2
def maxpool2(x):
    """F.max_pool2d(x, 2) (model_2.py:59-61,64). Returns (y, argmax) with argmax in {0,1,2,3}
    = first maximum in row-major window order (ATen keeps the first on ties)."""
    n, c, h, w = x.shape
    win = x.reshape(n, c, h // 2, 2, w // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    arg = win.argmax(axis=-1)
    y = np.take_along_axis(win, arg[..., None], axis=-1)[..., 0]
    return y, arg


def maxpool2_bwd(gy, arg):
    n, c, h2, w2 = gy.shape
    win = np.zeros((n, c, h2, w2, 4), dtype=gy.dtype)
    np.put_along_axis(win, arg[..., None], gy[..., None], axis=-1)
    return win.reshape(n, c, h2, w2, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h2 * 2, w2 * 2)


# --------------------------------------------------------------------------- conv transpose 2x2 s2
def convT2x2(x, w, b):
    """nn.ConvTranspose2d(Cin, Cout, 2, stride=2) (model_2.py:20,23,26,29); w is [Cin,Cout,2,2].

    out[n,co,2i+a,2j+b] = bias[co] + sum_ci x[n,ci,i,j] * w[ci,co,a,b]."""
    n, ci, h, wd = x.shape
    co = w.shape[1]
    out = np.zeros((n, co, 2 * h, 2 * wd), dtype=x.dtype)
    for a in range(2):
        for bb in range(2):
            out[:, :, a::2, bb::2] = np.einsum("nchw,co->nohw", x, w[:, :, a, bb], optimize=True)
    return out + b.reshape(1, co, 1, 1)


def convT2x2_bwd(x, w, gy):
    gx = np.zeros_like(x)
    gw = np.zeros_like(w)
    for a in range(2):
        for bb in range(2):
            g = gy[:, :, a::2, bb::2]
            gx += np.einsum("nohw,co->nchw", g, w[:, :, a, bb], optimize=True)
            gw[:, :, a, bb] = np.einsum("nchw,nohw->co", x, g, optimize=True)
    return gx, gw, gy.sum(axis=(0, 2, 3))


# --------------------------------------------------------------------------- head
def conv1x1(x, w, b):
    """nn.Conv2d(64, out_channels, 1) (model_2.py:32,79); w is [Cout,Cin,1,1]."""
    return np.einsum("oc,nchw->nohw", w[:, :, 0, 0], x, optimize=True) + b.reshape(1, -1, 1, 1)


def sigmoid(z):
    """torch.sigmoid (model_2.py:80)."""
    return (1.0 / (1.0 + np.exp(-z))).astype(z.dtype)


# --------------------------------------------------------------------------- loss
def focal_dice_loss(p, t, alpha=1.0, gamma=2.0, ratio=0.3, smooth=1e-7):
    """utils/metrics_DC.py:65-73 (+ FocalLoss.forward :43-63, dice_loss :11-17).

    F.binary_cross_entropy clamps each log term at -100."""
    logp = np.maximum(np.log(p), -100.0)
    log1mp = np.maximum(np.log1p(-p), -100.0)
    bce = -(t * logp + (1 - t) * log1mp)
    pt = np.exp(-bce)
    focal = (alpha * (1 - pt) ** gamma * bce).mean()
    inter = (p * t).sum(axis=(2, 3))
    union = p.sum(axis=(2, 3)) + t.sum(axis=(2, 3))
    dice = (2.0 * inter + smooth) / (union + smooth)
    return ratio * focal + (1 - ratio) * (1 - dice.mean())


def focal_dice_loss_bwd(p, t, alpha=1.0, gamma=2.0, ratio=0.3, smooth=1e-7):
    """d loss / d p of focal_dice_loss (analytic; clamp regions have zero derivative)."""
    logp = np.log(p)
    log1mp = np.log1p(-p)
    cl_p = logp > -100.0
    cl_q = log1mp > -100.0
    bce = -(t * np.maximum(logp, -100.0) + (1 - t) * np.maximum(log1mp, -100.0))
    dbce = -(t * cl_p / p - (1 - t) * cl_q / (1 - p))
    pt = np.exp(-bce)
    # d/dbce [ (1-pt)^g * bce ] = g (1-pt)^(g-1) * pt * bce + (1-pt)^g
    dfocal = alpha * (gamma * (1 - pt) ** (gamma - 1) * pt * bce + (1 - pt) ** gamma) * dbce / p.size
    inter = (p * t).sum(axis=(2, 3), keepdims=True)
    union = p.sum(axis=(2, 3), keepdims=True) + t.sum(axis=(2, 3), keepdims=True)
    nb = p.shape[0] * p.shape[1]
    ddice = (2.0 * t * (union + smooth) - (2.0 * inter + smooth)) / (union + smooth) ** 2
    return (ratio * dfocal - (1 - ratio) * ddice / nb).astype(p.dtype)


# --------------------------------------------------------------------------- whole network
class UNetOracle:
    """Restates UNetDC.forward (models/model_2.py:56-80) / UNet.forward (models/model.py:35-50)
    and its autograd, driven by a state-dict of numpy arrays with the reference's 136 keys."""

    def __init__(self, sd, dilations=None, dtype=np.float32):
        self.dt = dtype
        self.sd = {k: (np.asarray(v, dtype=dtype) if np.asarray(v).dtype.kind == "f" else np.asarray(v))
                   for k, v in sd.items()}
        self.dil = dict(DILATIONS_DC if dilations is None else dilations)
        self.new_running = {}

    # one conv -> BN -> ReLU stage
    def _stage(self, x, name, idx, d, train):
        sd = self.sd
        cw, cb = sd[f"{name}.{idx}.weight"], sd[f"{name}.{idx}.bias"]
        g, be = sd[f"{name}.{idx + 1}.weight"], sd[f"{name}.{idx + 1}.bias"]
        y = conv3x3(x, cw, cb, d)
        if train:
            n, cache = bn_train(y, g, be)
            cnt = y.shape[0] * y.shape[2] * y.shape[3]
            rm, rv = bn_running_update(sd[f"{name}.{idx + 1}.running_mean"],
                                       sd[f"{name}.{idx + 1}.running_var"], cache[2], cache[3], cnt)
            self.new_running[f"{name}.{idx + 1}.running_mean"] = rm
            self.new_running[f"{name}.{idx + 1}.running_var"] = rv
        else:
            n, cache = bn_eval(y, g, be, sd[f"{name}.{idx + 1}.running_mean"],
                               sd[f"{name}.{idx + 1}.running_var"]), None
        a = relu(n)
        self.tape.append(("stage", name, idx, d, x, cache, a))
        return a

    def _block(self, x, name, train):
        d = self.dil[name]
        return self._stage(self._stage(x, name, 0, d, train), name, 3, d, train)

    def forward(self, x, train=False, return_logits=False):
        x = np.asarray(x, dtype=self.dt)
        self.tape = []
        sd = self.sd
        skips = []
        h = x
        for name in ("enc1", "enc2", "enc3", "enc4"):
            h = self._block(h, name, train)
            skips.append(h)
            h, arg = maxpool2(h)
            self.tape.append(("pool", arg))
        h = self._block(h, "bottleneck", train)
        for lvl in (4, 3, 2, 1):
            up = convT2x2(h, sd[f"upconv{lvl}.weight"], sd[f"upconv{lvl}.bias"])
            self.tape.append(("up", lvl, h))
            skip = skips[lvl - 1]
            cat = np.concatenate([up, skip], axis=1)          # up-sampled first (model_2.py:68-77)
            self.tape.append(("cat", up.shape[1]))
            h = self._block(cat, f"dec{lvl}", train)
        z = conv1x1(h, sd["out_conv.weight"], sd["out_conv.bias"])
        p = sigmoid(z)
        self.tape.append(("head", h, p))
        self.z = z
        return (p, z) if return_logits else p

    def backward(self, gp):
        """Given dL/dprobs returns {param_key: grad}. Train-mode tape required."""
        sd = self.sd
        grads = {}
        tape = list(self.tape)
        kind, h, p = tape.pop()
        assert kind == "head"
        gz = gp * p * (1 - p)
        grads["out_conv.weight"] = np.einsum("nohw,nchw->oc", gz, h)[:, :, None, None]
        grads["out_conv.bias"] = gz.sum(axis=(0, 2, 3))
        g = np.einsum("oc,nohw->nchw", sd["out_conv.weight"][:, :, 0, 0], gz)
        skip_grads = {}
        pending_level = 1

        def stage_bwd(g, need_gx=True):
            kind, name, idx, d, x_in, cache, a = tape.pop()
            assert kind == "stage"
            g = g * (a > 0)
            g, gg, gb = bn_train_bwd(g, cache, sd[f"{name}.{idx + 1}.weight"])
            grads[f"{name}.{idx + 1}.weight"] = gg
            grads[f"{name}.{idx + 1}.bias"] = gb
            gx, gw, gcb = conv3x3_bwd(x_in, sd[f"{name}.{idx}.weight"], d, g, need_gx)
            grads[f"{name}.{idx}.weight"] = gw
            grads[f"{name}.{idx}.bias"] = gcb
            return gx

        for lvl in (1, 2, 3, 4):
            g = stage_bwd(stage_bwd(g))
            kind, cup = tape.pop()
            assert kind == "cat"
            skip_grads[lvl] = g[:, cup:]
            g = g[:, :cup]
            kind, l2, h_in = tape.pop()
            assert kind == "up" and l2 == lvl
            g, gw, gb = convT2x2_bwd(h_in, sd[f"upconv{lvl}.weight"], g)
            grads[f"upconv{lvl}.weight"] = gw
            grads[f"upconv{lvl}.bias"] = gb
        g = stage_bwd(stage_bwd(g))                       # bottleneck
        for lvl in (4, 3, 2, 1):
            kind, arg = tape.pop()
            assert kind == "pool"
            g = maxpool2_bwd(g, arg) + skip_grads[lvl]
            g = stage_bwd(g)
            g = stage_bwd(g, need_gx=(lvl != 1))
        assert not tape
        return grads
