"""Forward/backward schedule of the U-Net / U-Net-DC on the HIP kernels.

PyTorch is plumbing here: device memory (caching allocator), the current stream, autograd's
entry/exit points.  Every arithmetic op of ``UNetDC.forward`` (/root/reference/models/model_2.py:56-80)
and of its autograd is one of the C-ABI calls in ``include/unetdc_hip.h``.

Data layout in HBM
------------------
* activations: NHWC 2-D tensors ``[N*H*W, C]`` in the compute type (fp32 or bf16);
* decoder concat (``torch.cat([up, skip], 1)``, model_2.py:68-77): ONE ``[pixels, 2C]`` buffer per
  level -- the up-convolution writes columns ``[0, C)``, the encoder's normalise+ReLU pass writes
  its skip into columns ``[C, 2C)``; the concat itself moves no bytes, and in backward the two halves
  of the concat gradient are consumed in place (ConvT backward / encoder backward);
* per conv stage the raw (pre-BatchNorm) output ``y`` is kept for backward; the activation
  ``a = relu(scale*y + shift)`` is stored once (it is the next conv's input and the wgrad operand) --
  except where the consumer normalises ``y`` on load (training): the last stage (the head reads ``y``) and
  the first stage of the 64-channel blocks (the second stage's conv and weight gradient read ``y``, "bnin");
* parameters stay fp32 ``nn.Parameter``s in PyTorch layout; K-contiguous packed copies in the
  compute type are derived caches (ONE set per module, shared by the engines of every input shape)
  re-packed when a parameter's version counter changes or rewritten by the optimizer kernel;
* gradients are written by the kernels straight into one flat fp32 buffer in ``parameters()``
  order (so data-parallel buckets are contiguous slices, see dp.py).
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import call

_byref = ctypes.byref
# references to a tensor's storage (tensor + views + Python wrappers); absent on a torch build without it: allocate per step
_storage_use_count = getattr(torch._C, "_storage_Use_Count", None)

ENCODER = ("enc1", "enc2", "enc3", "enc4")
# fuse the BatchNorm-backward reduction into the dgrad epilogue that produces the gradient (A/B switch; the stand-alone
# reduction is also the fallback of the first-generation kernels)
FUSE_BN_BWD = os.environ.get("UNETDC_FUSE_BNBWD", "1") != "0"
# The head reads dec1's RAW conv output and applies that stage's BatchNorm + ReLU on load (unetdc_head_fwd_bn; the backward
# recomputes the activation the same way): dec1.3's normalisation pass and its 268 MB activation tensor disappear from the
# training step.  UNETDC_FUSE_HEAD_BN=0: stand-alone pass (A/B switch)
FUSE_HEAD_BN = os.environ.get("UNETDC_FUSE_HEAD_BN", "1") != "0" and FUSE_BN_BWD
# Second stage of a block fed from the first stage's RAW conv output ("bnin"): the consumer convolution and its weight gradient
# apply the first stage's BatchNorm + ReLU per staged tile in LDS (unetdc_conv3x3_fwd_bnin / unetdc_conv3x3_wgrad_bnin), so
# that stage's normalisation pass and activation tensor disappear.  Used where the library has the kernels (bf16, 64-channel
# blocks: enc1 and dec1, the two largest normalisation passes of the step).  UNETDC_FUSE_BNIN=0: stand-alone passes (A/B)
FUSE_BNIN = os.environ.get("UNETDC_FUSE_BNIN", "1") != "0"
# Settled in rounds 3-4 (bit-identical to the forms they replaced, which are gone from the schedule):
#  * the ConvTranspose2d bias gradient = column sums of the concat gradient, produced by the dgrad epilogue that writes it;
#  * one-channel head: the gradient of the head's input is dz * w[c] per pixel, so dec1's last stage recomputes it in its
#    BatchNorm-backward pass (unetdc_bn_relu_bwd_head) instead of reading a tensor the head backward wrote;
#  * first stage (enc1.0): its weight gradient applies the stage's BatchNorm + ReLU backward on load
#    (unetdc_conv3x3_first_wgrad_bn) -- nothing else reads that stage's dy unless dL/dx is asked for.
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class PackedWeights:
    """The K-contiguous compute-type weight images of ONE module (per device and compute type): ``w_fwd[9][Cout][Cin]`` /
    ``w_dgrad[9][Cin][Cout]`` per 3x3 conv (taps flipped), ``w_fwd[4*Cout][Cin]`` / ``w_dgrad[4][Cin][Cout]`` per
    ConvTranspose2d.  They depend on the parameters only, not on the input shape, so every engine of the module (one per
    input shape, unet.py) reads the same set and the optimizer kernel (optim.FusedAdam) writes one set."""
    _serials = __import__("itertools").count(1)

    def __init__(self, model, device, tdtype, dt):
        self.serial = next(PackedWeights._serials)     # identity that is never reused (id() of a freed object can be)
        self.model, self.device, self.tdtype, self.dt = model, device, tdtype, dt
        self.conv, self.up = {}, {}
        widths = [64, 128, 256, 512, 1024]
        prev = model.in_channels
        for l, name in enumerate(ENCODER + ("bottleneck",)):
            c = widths[l]
            if l > 0:                                   # the C_in = 1/3 first layer reads the fp32 master directly
                self._add_conv(name, 0, prev, c)
            self._add_conv(name, 3, c, c)
            prev = c
        for lvl in (4, 3, 2, 1):
            c = widths[lvl - 1]
            self._add_conv(f"dec{lvl}", 0, 2 * c, c)
            self._add_conv(f"dec{lvl}", 3, c, c)
            self.up[lvl] = dict(mod=getattr(model, f"upconv{lvl}"), cin=2 * c, cout=c,
                                w_fwd=torch.empty(4 * c * 2 * c, device=device, dtype=tdtype),
                                w_dgrad=torch.empty(4 * c * 2 * c, device=device, dtype=tdtype))
        self._versions = None          # parameter version counters the images were last built from
        self._fresh_versions = None    # set by an optimizer that has just written the images itself
        self._ptrs = None
        self._trained = False          # a train-mode forward has run: optimizer steps may follow at any time

    def _add_conv(self, block, idx, cin, cout):
        conv = getattr(self.model, block)[idx]
        self.conv[(block, idx)] = dict(mod=conv, cin=cin, cout=cout,
                                       w_fwd=torch.empty(9 * cout * cin, device=self.device, dtype=self.tdtype),
                                       w_dgrad=torch.empty(9 * cout * cin, device=self.device, dtype=self.tdtype))

    def entries(self):
        """(parameter, fwd image, dgrad image, a, b, kind) for every packed tensor, in a fixed order."""
        ent = [(c["mod"].weight, c["w_fwd"], c["w_dgrad"], c["cout"], c["cin"], 0) for c in self.conv.values()]
        ent += [(u["mod"].weight, u["w_fwd"], u["w_dgrad"], u["cin"], u["cout"], 1) for u in self.up.values()]
        return ent

    def invalidate(self):
        """Force the next forward to rebuild the images (call after writing parameters through a path that bumps no
        version counter)."""
        self._versions = None

    def fresh(self, versions):
        """Called by an optimizer that writes the images itself (FusedAdam, optim.py) with the version counters of the
        packed parameters at that moment: the next forward needs no re-pack IF those counters still stand (any write in
        between -- load_state_dict, copy_, clamp_, an EMA swap -- bumps one of them and the images are rebuilt)."""
        self._fresh_versions = tuple(versions)

    def ensure(self, need_dgrad):
        """Re-pack the images when they may be stale: ONE launch over a device-resident descriptor table.  Fused
        optimizers such as torch.optim.Adam(fused=True) update parameters WITHOUT bumping their version counters, so
        once a train-mode forward has run on this module (an optimizer may be stepping the parameters) the images are
        rebuilt on every forward, eval-mode ones included (train fwd, eval fwd, opt.step(), eval fwd must not see stale
        weights; the launch costs ~0.1 ms); in a pure-inference process the version counters (load_state_dict, copy_)
        decide.  An optimizer that writes the images itself (optim.FusedAdam) announces it through fresh() and the pack
        is skipped while the counters it saw still stand."""
        import numpy as np
        ent = self.entries()
        versions = tuple(w._version for w, *_ in ent)
        ptrs = tuple(w.data_ptr() for w, *_ in ent)
        if self._ptrs != ptrs:
            dt = np.dtype([("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("begin", "<i8"), ("a", "<i4"), ("b", "<i4"),
                           ("kind", "<i4"), ("pad", "<i4")])
            tab = np.zeros(len(ent), dtype=dt)
            off = 0
            for i, (w, wf, wd, a, b, kind) in enumerate(ent):
                tab[i] = (w.data_ptr(), wf.data_ptr(), wd.data_ptr(), off, a, b, kind, 0)
                off += (a // 32) * (b // 32)                 # 32 x 32 channel tiles of this tensor
            self._table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)
            self._total, self._ptrs, self._versions = off, ptrs, None
        self._trained = self._trained or need_dgrad
        fresh, self._fresh_versions = self._fresh_versions, None
        if fresh is not None and fresh == versions and self._versions is not None:
            self._versions = versions          # the optimizer step wrote both images and nothing touched the parameters since
            return
        if self._trained or self._versions != versions:
            call("unetdc_pack_many", self._table.data_ptr(), len(ent), self._total, self.dt, _stream())
            self._versions = versions


class _Stage:
    """One conv3x3 -> BatchNorm -> ReLU stage: parameters, packed weights, saved tensors."""

    def __init__(self, eng, block, idx, cin, cout, dil, npix, hw, first=False):
        m = getattr(eng.model, block)
        self.name = f"{block}.{idx}"
        self.conv, self.bn = m[idx], m[idx + 1]
        self.cin, self.cout, self.dil, self.npix, self.hw, self.first = cin, cout, dil, npix, hw, first
        dev, dt = eng.device, eng.tdtype
        f32 = dict(device=dev, dtype=torch.float32)
        self.y = torch.empty(npix, cout, device=dev, dtype=dt)          # raw conv output (pre-BN)
        if first:
            rows = _lib.load().unetdc_conv3x3_first_stats_rows(npix, cin, cout)
        else:
            rows = _lib.load().unetdc_conv3x3_stats_rows(npix, cout)
            img = eng.weights.conv[(block, idx)]
            self.w_fwd, self.w_dgrad = img["w_fwd"], img["w_dgrad"]
        self.stat_rows = rows
        self.stats = torch.empty((rows + 64) * 2 * cout, **f32)
        self.scale, self.shift = torch.empty(cout, **f32), torch.empty(cout, **f32)
        self.mean, self.rstd = torch.empty(cout, **f32), torch.empty(cout, **f32)
        self.bnin = None      # (scale, shift) of the stage whose RAW output is this stage's input (normalised on load)
        self.x_in = None      # input view of the last forward (for wgrad)
        self.a_out = None     # activated output view
        # BatchNorm-backward partial sums produced by the dgrad kernel that writes this stage's
        # incoming gradient (fused reduction); rows = 256-pixel blocks, 64 spare rows for the 2nd stage
        self.bwd_rows = _lib.load().unetdc_conv3x3_stats_rows(npix, cout)
        self.bwd_parts = None
        self.bwd_nparts = 0
        self.dy = None        # gradient of this stage's conv output


class UNetEngine:
    def __init__(self, model, x, weights=None):
        _lib.load()
        if not x.is_cuda:
            raise _lib.UnetdcError("UNetEngine needs a HIP device tensor")
        self.model = model
        self.device = x.device
        self.N, self.cin, self.H, self.W = x.shape
        if self.H % 16 or self.W % 16:
            raise ValueError(f"H and W must be multiples of 16 (4 poolings), got {self.H}x{self.W}")
        if self.cin != model.in_channels:
            raise ValueError(f"expected {model.in_channels} input channels, got {self.cin}")
        self.dtype_name = model.compute_dtype
        self.dt = _lib.BF16 if self.dtype_name == "bf16" else _lib.F32
        self.tdtype = torch.bfloat16 if self.dtype_name == "bf16" else torch.float32
        self.oc = model.out_channels
        # packed weight images: shared by every engine of the module (they do not depend on the input shape)
        if weights is None or weights.device != self.device or weights.dt != self.dt or weights.model is not model:
            weights = PackedWeights(model, self.device, self.tdtype, self.dt)
        self.weights = weights
        self._build()

    # ------------------------------------------------------------------ construction
    def matches(self, x):
        return (x.device == self.device and tuple(x.shape) == (self.N, self.cin, self.H, self.W)
                and self.dtype_name == self.model.compute_dtype)

    def _build(self):
        dev, dt = self.device, self.tdtype
        N, H, W = self.N, self.H, self.W
        widths = [64, 128, 256, 512, 1024]
        self.res = [(H >> l, W >> l) for l in range(5)]
        self.npix = [N * h * w for h, w in self.res]
        d = self.model.DILATIONS
        lib = _lib.load()
        self.stages = {}
        prev = self.cin
        for l, name in enumerate(ENCODER + ("bottleneck",)):
            c = widths[l]
            self.stages[(name, 0)] = _Stage(self, name, 0, prev, c, d[name], self.npix[l], self.res[l],
                                            first=(l == 0))
            self.stages[(name, 3)] = _Stage(self, name, 3, c, c, d[name], self.npix[l], self.res[l])
            prev = c
        for lvl in (4, 3, 2, 1):
            c = widths[lvl - 1]
            name = f"dec{lvl}"
            self.stages[(name, 0)] = _Stage(self, name, 0, 2 * c, c, d[name], self.npix[lvl - 1], self.res[lvl - 1])
            self.stages[(name, 3)] = _Stage(self, name, 3, c, c, d[name], self.npix[lvl - 1], self.res[lvl - 1])
        # activations
        self.a0 = {}      # activated output of stage 0 of each block
        self.a3 = {}      # activated output of stage 3 for bottleneck / decoder blocks
        self.cat = {}     # concat buffers per level 1..4
        self.pool = {}    # pooled encoder outputs per level 1..4
        for l, name in enumerate(ENCODER):
            c = widths[l]
            self.a0[name] = torch.empty(self.npix[l], c, device=dev, dtype=dt)
            self.cat[l + 1] = torch.empty(self.npix[l], 2 * c, device=dev, dtype=dt)
            self.pool[l + 1] = torch.empty(self.npix[l + 1], c, device=dev, dtype=dt)
        self.a0["bottleneck"] = torch.empty(self.npix[4], 1024, device=dev, dtype=dt)
        self.a3["bottleneck"] = torch.empty(self.npix[4], 1024, device=dev, dtype=dt)
        for lvl in (4, 3, 2, 1):
            c = widths[lvl - 1]
            self.a0[f"dec{lvl}"] = torch.empty(self.npix[lvl - 1], c, device=dev, dtype=dt)
            self.a3[f"dec{lvl}"] = torch.empty(self.npix[lvl - 1], c, device=dev, dtype=dt)
        # transposed convs: module, sizes, packed weights (shared), input view of the last forward
        self.up = {lvl: dict(self.weights.up[lvl]) for lvl in (4, 3, 2, 1)}
        # blocks whose second stage normalises the first stage's raw output on load
        # (value 1: the activation of stage 0 is never stored, the weight gradient normalises on load too; 2: the second stage's
        #  forward stores it as a by-product and the plain weight-gradient kernel reads it -- unetdc_conv3x3_bnin_supported)
        self.bnin_blocks = {}
        if FUSE_BNIN:
            for (name, idx), st in self.stages.items():
                h, w = st.hw
                mode = lib.unetdc_conv3x3_bnin_supported(N, h, w, st.cin, st.cout, st.dil, self.dt) if idx == 3 else 0
                if mode:
                    self.bnin_blocks[name] = mode
        # gradient-side buffers (allocated lazily on the first backward)
        self.grad_bufs = None
        # parameter order == model.parameters() order; flat gradient offsets
        self.params = list(self.model.parameters())
        self.pindex = {id(p): i for i, p in enumerate(self.params)}
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += p.numel()
        self.poffs, self.nparams = offs, o
        # workspace: the largest request of any backward op
        need = 1 << 20
        for (name, idx), st in self.stages.items():
            h, w = st.hw
            if st.first:
                need = max(need, lib.unetdc_conv3x3_first_wgrad_workspace(N, h, w, st.cin, st.cout))
            else:
                need = max(need, lib.unetdc_conv3x3_wgrad_workspace(N, h, w, st.cin, st.cout, self.dt))
            need = max(need, lib.unetdc_bn_relu_bwd_workspace(N, h, w, st.cout, 0, self.dt))
            need = max(need, lib.unetdc_bn_relu_bwd_workspace(N, h, w, st.cout, 1, self.dt))
        for lvl, u in self.up.items():
            h, w = self.res[lvl]
            need = max(need, lib.unetdc_convT2x2_wgrad_workspace(N, h, w, u["cin"], u["cout"], self.dt))
            need = max(need, lib.unetdc_channel_sum_workspace(self.npix[lvl - 1], u["cout"]))
            need = max(need, lib.unetdc_conv3x3_dgrad_colsum_workspace(N, 2 * h, 2 * w, 2 * u["cout"]))
        need = max(need, lib.unetdc_head_bwd_workspace(N, H, W, 64, self.oc, self.dt))
        self.ws_bytes = int(need)
        self.workspace = None
        # Saved-for-backward activations live in this engine's buffers (one set, sized for 288 GB of HBM, nothing is
        # recomputed); `generation` counts forwards so that a backward can tell whether ITS forward's activations
        # are still the ones in the buffers (see _UNetFunction.backward).
        self.generation = 0
        # busy: the activations in the buffers belong to a live autograd graph (set by _UNetFunction.forward, cleared when its
        # backward has run or its graph has been freed); the module gives another forward of this shape its own engine meanwhile
        self.busy = False
        self._busy_gen = -1
        self._rows, self._np = ctypes.c_int(0), ctypes.c_int(0)      # out-parameters of the C ABI (statistics rows that carry data)
        self._flat = None              # flat fp32 gradient buffer, kept across steps (see _flat_grads)

    # ------------------------------------------------------------------ weight caches
    def invalidate_weight_cache(self):
        self.weights.invalidate()

    # ------------------------------------------------------------------ forward
    def run(self, x):
        """Called by the nn.Module: returns probabilities [N, OC, H, W] fp32 (autograd aware)."""
        model = self.model
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.params))
        if model.training and not needs_grad:
            # train-mode BatchNorm under no_grad: same arithmetic, nothing saved
            return self.forward(x, train=True)
        if needs_grad:
            # eval mode with gradients enabled (fine-tuning with frozen BatchNorm statistics): the training-path kernels with
            # mean / variance taken from the running buffers, which stay untouched
            return _UNetFunction.apply(x, self, not model.training, *self.params)
        return self.forward(x, train=False)

    def _stage_fwd(self, st, xin, dst, train, pooled=None, apply=True, frozen=False, bnin=None, act_out=None):
        """conv -> BN -> ReLU.  xin: [npix, cin] view (or the NCHW image for the first stage);
        dst: [npix, cout] view receiving the activation; pooled: optional [npix/4, cout] view;
        apply=False (train mode only): stop after the batch statistics -- the consumer normalises on load."""
        s = _stream()
        N = self.N
        h, w = st.hw
        conv, bn = st.conv, st.bn
        st.x_in, st.a_out = xin, dst
        st.bnin = bnin if train else None
        if train:
            y = st.y
            if bnin is not None:                   # xin = the RAW output of the stage in front, normalised per staged patch
                call("unetdc_conv3x3_fwd_bnin", xin.data_ptr(), xin.stride(0), bnin[0].data_ptr(), bnin[1].data_ptr(),
                     st.w_fwd.data_ptr(), conv.bias.data_ptr(), y.data_ptr(), y.stride(0), st.stats.data_ptr(), _byref(self._rows),
                     _ptr(act_out), act_out.stride(0) if act_out is not None else 0, N, h, w, st.cin, st.cout, st.dil, self.dt, s)
                st.stat_rows = self._rows.value
                if act_out is not None:            # the forward stored the normalised input: the weight gradient reads it
                    st.x_in, st.bnin = act_out, None
            elif st.first:
                call("unetdc_conv3x3_first_fwd", xin.data_ptr(), conv.weight.data_ptr(), conv.bias.data_ptr(),
                     None, None, y.data_ptr(), y.stride(0), st.stats.data_ptr(), N, h, w, st.cin, st.cout,
                     st.dil, self.dt, s)
            else:
                call("unetdc_conv3x3_fwd", xin.data_ptr(), xin.stride(0), st.w_fwd.data_ptr(), conv.bias.data_ptr(),
                     None, None, y.data_ptr(), y.stride(0), st.stats.data_ptr(), _byref(self._rows), N, h, w, st.cin,
                     st.cout, st.dil, self.dt, s)
                st.stat_rows = self._rows.value                         # rows that carry data (<= the sizing bound)
            track = bn.track_running_stats and bn.running_mean is not None
            mom = BN_MOMENTUM if bn.momentum is None else bn.momentum
            if frozen and track:
                call("unetdc_bn_frozen_affine", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                     bn.running_var.data_ptr(), bn.eps, st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(),
                     st.rstd.data_ptr(), st.cout, s)
            else:                                              # (eval mode without running buffers = batch statistics, as nn.BatchNorm2d)
                call("unetdc_bn_finalize", st.stats.data_ptr(), st.stat_rows, st.npix, bn.weight.data_ptr(),
                     bn.bias.data_ptr(), bn.eps, mom, _ptr(bn.running_mean) if track and not frozen else None,
                     _ptr(bn.running_var) if track and not frozen else None, st.scale.data_ptr(), st.shift.data_ptr(),
                     st.mean.data_ptr(), st.rstd.data_ptr(), st.cout, s)
            if track and not frozen:
                self._nbt.append(bn.num_batches_tracked)       # incremented together at the end of forward()
            if apply:
                call("unetdc_bn_relu_apply", y.data_ptr(), y.stride(0), st.scale.data_ptr(), st.shift.data_ptr(),
                     dst.data_ptr(), dst.stride(0), _ptr(pooled), pooled.stride(0) if pooled is not None else 0,
                     N, h, w, st.cout, self.dt, s)
        else:
            call("unetdc_bn_eval_affine", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                 bn.running_var.data_ptr(), conv.bias.data_ptr(), bn.eps, st.scale.data_ptr(), st.shift.data_ptr(),
                 st.cout, s)
            if st.first:
                call("unetdc_conv3x3_first_fwd", xin.data_ptr(), conv.weight.data_ptr(), None, st.scale.data_ptr(),
                     st.shift.data_ptr(), dst.data_ptr(), dst.stride(0), None, N, h, w, st.cin, st.cout, st.dil,
                     self.dt, s)
            else:
                call("unetdc_conv3x3_fwd", xin.data_ptr(), xin.stride(0), st.w_fwd.data_ptr(), None,
                     st.scale.data_ptr(), st.shift.data_ptr(), dst.data_ptr(), dst.stride(0), None, None, N, h, w,
                     st.cin, st.cout, st.dil, self.dt, s)
            if pooled is not None:
                call("unetdc_bn_relu_apply", dst.data_ptr(), dst.stride(0), None, None, None, 0,
                     pooled.data_ptr(), pooled.stride(0), N, h, w, st.cout, self.dt, s)

    def forward(self, x, train, frozen=False):
        """train: the training-path kernels (batch statistics, everything saved for backward); frozen (with train): BatchNorm
        statistics from the running buffers instead of the batch (eval mode under autograd)."""
        self.generation += 1               # every forward overwrites the activation buffers
        self._frozen = bool(frozen)
        self.weights.ensure(need_dgrad=train)
        self._nbt = []
        s = _stream()
        N = self.N
        widths = [64, 128, 256, 512, 1024]
        hin = x
        for l, name in enumerate(ENCODER):
            c = widths[l]
            s0 = self.stages[(name, 0)]
            fuse = train and name in self.bnin_blocks      # stage 3 reads stage 0's raw output and normalises it on load
            self._stage_fwd(s0, hin, self.a0[name], train, frozen=frozen, apply=not fuse)
            skip = self.cat[l + 1][:, c:]
            self._stage_fwd(self.stages[(name, 3)], s0.y if fuse else self.a0[name], skip, train, pooled=self.pool[l + 1],
                            frozen=frozen, bnin=(s0.scale, s0.shift) if fuse else None,
                            act_out=self.a0[name] if fuse and self.bnin_blocks[name] == 2 else None)
            hin = self.pool[l + 1]
        self._stage_fwd(self.stages[("bottleneck", 0)], hin, self.a0["bottleneck"], train, frozen=frozen)
        self._stage_fwd(self.stages[("bottleneck", 3)], self.a0["bottleneck"], self.a3["bottleneck"], train, frozen=frozen)
        hin = self.a3["bottleneck"]
        for lvl in (4, 3, 2, 1):
            u = self.up[lvl]
            c = u["cout"]
            h, w = self.res[lvl]                               # input resolution of the up-conv
            upv = self.cat[lvl][:, :c]
            u["x_in"] = hin
            call("unetdc_convT2x2_fwd", hin.data_ptr(), hin.stride(0), u["w_fwd"].data_ptr(),
                 u["mod"].bias.data_ptr(), upv.data_ptr(), upv.stride(0), N, h, w, u["cin"], c, self.dt, s)
            name = f"dec{lvl}"
            head_norm = train and FUSE_HEAD_BN and lvl == 1          # dec1.3: normalised by the head while loading
            s0 = self.stages[(name, 0)]
            fuse = train and name in self.bnin_blocks
            self._stage_fwd(s0, self.cat[lvl], self.a0[name], train, frozen=frozen, apply=not fuse)
            self._stage_fwd(self.stages[(name, 3)], s0.y if fuse else self.a0[name], self.a3[name], train,
                            apply=not head_norm, frozen=frozen, bnin=(s0.scale, s0.shift) if fuse else None,
                            act_out=self.a0[name] if fuse and self.bnin_blocks[name] == 2 else None)
            hin = self.a3[name]
        probs = torch.empty(N, self.oc, self.H, self.W, device=self.device, dtype=torch.float32)
        oc = self.model.out_conv
        if train and FUSE_HEAD_BN:
            last = self.stages[("dec1", 3)]
            call("unetdc_head_fwd_bn", last.y.data_ptr(), last.y.stride(0), last.scale.data_ptr(), last.shift.data_ptr(),
                 oc.weight.data_ptr(), oc.bias.data_ptr(), probs.data_ptr(), N, self.H, self.W, 64, self.oc, self.dt, s)
            hin = None                                              # no activation tensor: the backward recomputes it from y
        else:
            call("unetdc_head_fwd", hin.data_ptr(), hin.stride(0), oc.weight.data_ptr(), oc.bias.data_ptr(),
                 probs.data_ptr(), N, self.H, self.W, 64, self.oc, self.dt, s)
        self.head_in = hin
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)                  # nn.BatchNorm2d's num_batches_tracked += 1, one launch
        return probs

    # ------------------------------------------------------------------ backward
    def _ensure_grad_bufs(self):
        if self.grad_bufs is not None:
            return
        dev, dt = self.device, self.tdtype
        g = {}
        widths = [64, 128, 256, 512, 1024]
        for l in range(5):
            c = widths[l]
            g[("da", l)] = torch.empty(self.npix[l], c, device=dev, dtype=dt)       # grad of an activation
            if l < 4:
                g[("dcat", l + 1)] = torch.empty(self.npix[l], 2 * c, device=dev, dtype=dt)
                g[("dpool", l + 1)] = torch.empty(self.npix[l + 1], c, device=dev, dtype=dt)
        for st in self.stages.values():
            st.dy = torch.empty(st.npix, st.cout, device=dev, dtype=dt)
        self.grad_bufs = g
        self.workspace = torch.empty(self.ws_bytes, device=dev, dtype=torch.uint8)

    def _flat_grads(self):
        """The flat fp32 gradient buffer of this backward (124 MB for the U-Net-DC).  ONE buffer is kept per engine and handed
        out again when nothing but the engine still refers to its storage -- the usual training loop, where zero_grad() has
        dropped the previous step's ``.grad`` views (train_DC_focal.py:251) -- so that a step never goes back to the caching
        allocator for its largest block.  While anything still views it (gradient accumulation without zero_grad, gradients
        kept by the caller, torch.autograd.grad results) a fresh buffer is allocated instead and becomes the kept one:
        gradients handed out are never overwritten by a later backward."""
        f = self._flat
        if f is not None and _storage_use_count is not None and _storage_use_count(f.untyped_storage()._cdata) <= 2:
            return f                   # 2 = this tensor + the Python storage wrapper of the query
        self._flat = torch.empty(self.nparams, device=self.device, dtype=torch.float32)
        return self._flat

    def _gview(self, flat, p):
        i = self.pindex[id(p)]
        return flat[self.poffs[i]: self.poffs[i] + p.numel()]

    def _bnstats_args(self, prev):
        """Arguments describing the stage whose BN-backward reduction a dgrad epilogue should fuse."""
        if prev.bwd_parts is None:
            prev.bwd_parts = torch.empty((prev.bwd_rows + 64) * 3 * prev.cout, device=self.device, dtype=torch.float32)
        return (prev.y.data_ptr(), prev.y.stride(0), prev.scale.data_ptr(), prev.shift.data_ptr(),
                prev.mean.data_ptr(), prev.rstd.data_ptr(), prev.bwd_parts.data_ptr(), prev.bwd_parts.numel(),
                _byref(self._np))

    def _stage_bwd(self, st, flat, lvl, dskip, dpool, dx_out, fuse_prev=None, colsum=None, head=None):
        """Backward of one stage.  dskip/dpool: incoming gradient(s) of the activation;
        dx_out: [npix, cin] view to receive the input gradient (None for the first stage);
        fuse_prev: the stage consuming dx_out as its activation gradient -- its BatchNorm-backward
        reduction is then fused into this stage's dgrad epilogue;
        colsum: (fp32 out, c0, c) -- per-channel sums of dx_out[:, c0:c0+c] produced by the dgrad epilogue
        (the ConvTranspose2d bias gradient when dx_out is the gradient of the concat buffer)."""
        s = _stream()
        N = self.N
        h, w = st.hw
        dy = st.dy
        ws, wsb = self.workspace.data_ptr(), self.ws_bytes
        pre = (st.bwd_parts.data_ptr(), st.bwd_nparts) if (st.bwd_nparts and dpool is None) else (None, 0)
        if (st.first and pre[0] is not None and dskip is not None and not self._frozen
              and not getattr(self, "_need_dx", False)
              and _lib.load().unetdc_conv3x3_first_wgrad_bn_supported(N, h, w, st.cin, st.cout, st.dil, self.dt)):
            # BatchNorm backward of the first stage on load of its weight gradient: dy is never written
            if getattr(st, "bwd_coeffs", None) is None:
                st.bwd_coeffs = torch.empty(3 * st.cout, device=self.device, dtype=torch.float32)
            call("unetdc_bn_relu_bwd_coeffs", pre[0], pre[1], st.bn.weight.data_ptr(), st.rstd.data_ptr(),
                 self._gview(flat, st.bn.weight).data_ptr(), self._gview(flat, st.bn.bias).data_ptr(),
                 self._gview(flat, st.conv.bias).data_ptr(), st.bwd_coeffs.data_ptr(), N, h, w, st.cout, s)
            call("unetdc_conv3x3_first_wgrad_bn", st.x_in.data_ptr(), dskip.data_ptr(), dskip.stride(0), st.y.data_ptr(),
                 st.y.stride(0), st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(),
                 st.bwd_coeffs.data_ptr(), self._gview(flat, st.conv.weight).data_ptr(), ws, wsb, N, h, w, st.cin, st.cout,
                 st.dil, self.dt, s)
            st.bwd_nparts = 0
            return
        elif head is not None:
            # (dprobs, probs, head weight): the incoming gradient is recomputed per pixel, `dskip` was never written
            call("unetdc_bn_relu_bwd_head", head[0].data_ptr(), head[1].data_ptr(), head[2].data_ptr(), st.y.data_ptr(),
                 st.y.stride(0), st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(),
                 st.bn.weight.data_ptr(), dy.data_ptr(), dy.stride(0), self._gview(flat, st.bn.weight).data_ptr(),
                 self._gview(flat, st.bn.bias).data_ptr(), self._gview(flat, st.conv.bias).data_ptr(), ws, wsb,
                 pre[0], pre[1], N, h, w, st.cout, self.dt, s)
        else:
            self._bn_relu_bwd_plain(st, flat, dskip, dpool, dy, pre, ws, wsb, N, h, w, s)
        st.bwd_nparts = 0
        self._stage_bwd_rest(st, flat, lvl, dx_out, fuse_prev, colsum, dy, ws, wsb, N, h, w, s)

    def _bn_relu_bwd_plain(self, st, flat, dskip, dpool, dy, pre, ws, wsb, N, h, w, s):
        call("unetdc_bn_relu_bwd_frozen" if self._frozen and st.bn.running_mean is not None else "unetdc_bn_relu_bwd", _ptr(dskip), dskip.stride(0) if dskip is not None else 0,
             _ptr(dpool), dpool.stride(0) if dpool is not None else 0, st.y.data_ptr(), st.y.stride(0),
             st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(),
             st.bn.weight.data_ptr(), dy.data_ptr(), dy.stride(0), self._gview(flat, st.bn.weight).data_ptr(),
             self._gview(flat, st.bn.bias).data_ptr(), self._gview(flat, st.conv.bias).data_ptr(), ws, wsb,
             pre[0], pre[1], N, h, w, st.cout, self.dt, s)

    def _stage_bwd_rest(self, st, flat, lvl, dx_out, fuse_prev, colsum, dy, ws, wsb, N, h, w, s):
        dw = self._gview(flat, st.conv.weight)
        xin = st.x_in

        def wgrad():
            s2, ws2 = s, ws
            if st.first:
                call("unetdc_conv3x3_first_wgrad", xin.data_ptr(), dy.data_ptr(), dy.stride(0), dw.data_ptr(), ws2, wsb,
                     N, h, w, st.cin, st.cout, st.dil, self.dt, s2)
            elif st.bnin is not None:
                call("unetdc_conv3x3_wgrad_bnin", xin.data_ptr(), xin.stride(0), st.bnin[0].data_ptr(),
                     st.bnin[1].data_ptr(), dy.data_ptr(), dy.stride(0), dw.data_ptr(), ws2, wsb, N, h, w, st.cin,
                     st.cout, st.dil, self.dt, s2)
            else:
                call("unetdc_conv3x3_wgrad", xin.data_ptr(), xin.stride(0), dy.data_ptr(), dy.stride(0), dw.data_ptr(),
                     ws2, wsb, N, h, w, st.cin, st.cout, st.dil, self.dt, s2)

        wgrad()
        self._stage_dgrad(st, dx_out, fuse_prev, colsum, dy, ws, wsb, N, h, w, s)

    def _stage_dgrad(self, st, dx_out, fuse_prev, colsum, dy, ws, wsb, N, h, w, s):
        if not st.first:
            if dx_out is not None and fuse_prev is not None and FUSE_BN_BWD:
                call("unetdc_conv3x3_dgrad_bnstats", dy.data_ptr(), dy.stride(0), st.w_dgrad.data_ptr(),
                     dx_out.data_ptr(), dx_out.stride(0), *self._bnstats_args(fuse_prev), N, h, w, st.cin, st.cout,
                     st.dil, self.dt, s)
                fuse_prev.bwd_nparts = self._np.value
            elif dx_out is not None and colsum is not None:
                call("unetdc_conv3x3_dgrad_colsum", dy.data_ptr(), dy.stride(0), st.w_dgrad.data_ptr(), dx_out.data_ptr(),
                     dx_out.stride(0), colsum[0].data_ptr(), colsum[1], colsum[2], ws, wsb, N, h, w, st.cin, st.cout,
                     st.dil, self.dt, s)
            elif dx_out is not None:
                call("unetdc_conv3x3_dgrad", dy.data_ptr(), dy.stride(0), st.w_dgrad.data_ptr(), dx_out.data_ptr(),
                     dx_out.stride(0), N, h, w, st.cin, st.cout, st.dil, self.dt, s)

    def _block_bwd(self, name, flat, lvl, dskip, dpool, dx_out, colsum=None, head=None):
        """stage 3 then stage 0 of a block; the gradient between them lives in the 'da' buffer."""
        da = self.grad_bufs[("da", lvl)]
        s3, s0 = self.stages[(name, 3)], self.stages[(name, 0)]
        self._stage_bwd(s3, flat, lvl, dskip, dpool, da, fuse_prev=s0, head=head)
        self._notify(flat, [s3.conv, s3.bn])             # per STAGE: bottleneck.3's 37.7 MB travel while bottleneck.0 computes
        self._stage_bwd(s0, flat, lvl, da, None, dx_out, colsum=colsum)
        self._notify(flat, [s0.conv, s0.bn])

    def _notify(self, flat, mods):
        """Tell the data-parallel wrapper that the gradients of `mods` (adjacent in parameters() order) are enqueued."""
        hook = self.model.grad_ready_hook
        if hook is not None:
            ps = [q for m in mods for q in m.parameters()]
            lo = self.poffs[self.pindex[id(ps[0])]]
            hi = self.poffs[self.pindex[id(ps[-1])]] + ps[-1].numel()
            assert hi - lo == sum(q.numel() for q in ps), "gradient ranges must be contiguous in the flat buffer"
            hook(flat, lo, hi)

    def input_grad(self):
        """dL/dx [N, C_in, H, W] fp32 of the backward that has just been enqueued: the first convolution's dgrad from the
        gradient of its output (kept in the stage's dy buffer).  Only computed when the input requires a gradient."""
        st = self.stages[("enc1", 0)]
        dx = torch.empty(self.N, self.cin, self.H, self.W, device=self.device, dtype=torch.float32)
        call("unetdc_conv3x3_first_dgrad", st.dy.data_ptr(), st.dy.stride(0), st.conv.weight.data_ptr(), dx.data_ptr(), self.N,
             self.H, self.W, self.cin, st.cout, st.dil, self.dt, _stream())
        return dx

    def backward(self, dprobs, probs, need_dx=False):
        """dprobs, probs: [N, OC, H, W] fp32.  Returns the flat fp32 gradient buffer (parameters() order).
        need_dx: input_grad() will be called afterwards (the first stage then keeps its dy)."""
        self._ensure_grad_bufs()
        self._need_dx = bool(need_dx)
        s = _stream()
        N = self.N
        g = self.grad_bufs
        ws, wsb = self.workspace.data_ptr(), self.ws_bytes
        flat = self._flat_grads()
        dprobs = dprobs.contiguous()
        oc = self.model.out_conv
        da = g[("da", 0)]
        last = self.stages[("dec1", 3)]                  # its activated output feeds out_conv
        # one output channel, training statistics: the head's input gradient is never stored (dec1.3 recomputes it)
        head = (dprobs, probs, oc.weight) if (FUSE_BN_BWD and self.oc == 1 and not self._frozen) else None
        if FUSE_BN_BWD:                                  # da's BatchNorm-backward sums come out of the same pass
            call("unetdc_head_bwd_bnstats", dprobs.data_ptr(), probs.data_ptr(), _ptr(self.head_in),
                 self.head_in.stride(0) if self.head_in is not None else 64, oc.weight.data_ptr(),
                 None if head is not None else da.data_ptr(), da.stride(0),
                 self._gview(flat, oc.weight).data_ptr(), self._gview(flat, oc.bias).data_ptr(), ws, wsb,
                 *self._bnstats_args(last), N, self.H, self.W, 64, self.oc, self.dt, s)
            last.bwd_nparts = self._np.value
        else:
            call("unetdc_head_bwd", dprobs.data_ptr(), probs.data_ptr(), self.head_in.data_ptr(),
                 self.head_in.stride(0), oc.weight.data_ptr(), da.data_ptr(), da.stride(0),
                 self._gview(flat, oc.weight).data_ptr(), self._gview(flat, oc.bias).data_ptr(), ws, wsb,
                 N, self.H, self.W, 64, self.oc, self.dt, s)
        self._notify(flat, [self.model.out_conv])
        # decoder, level 1 (full resolution) up to level 4
        dact = da                               # gradient of the current block's activated output
        for lvl in (1, 2, 3, 4):
            l = lvl - 1
            u = self.up[lvl]
            c = u["cout"]
            dcat = g[("dcat", lvl)]
            # the dgrad that writes dcat = grad of cat([up, enc]) also sums its first half per channel = upconv bias grad
            self._block_bwd(f"dec{lvl}", flat, l, dact, None, dcat, colsum=(self._gview(flat, u["mod"].bias), 0, c),
                            head=head if lvl == 1 else None)
            dup = dcat[:, :c]
            h, w = self.res[lvl]
            xin = u["x_in"]
            call("unetdc_convT2x2_wgrad", xin.data_ptr(), xin.stride(0), dup.data_ptr(), dup.stride(0),
                 self._gview(flat, u["mod"].weight).data_ptr(), ws, wsb, N, h, w, u["cin"], c, self.dt, s)
            dnext = g[("da", lvl)]               # gradient w.r.t. the up-conv input (level lvl+1 resolution)
            prev = self.stages[("bottleneck" if lvl == 4 else f"dec{lvl + 1}", 3)]     # producer of the up-conv input
            if FUSE_BN_BWD:
                call("unetdc_convT2x2_dgrad_bnstats", dup.data_ptr(), dup.stride(0), u["w_dgrad"].data_ptr(),
                     dnext.data_ptr(), dnext.stride(0), *self._bnstats_args(prev), N, h, w, u["cin"], c, self.dt, s)
                prev.bwd_nparts = self._np.value
            else:
                call("unetdc_convT2x2_dgrad", dup.data_ptr(), dup.stride(0), u["w_dgrad"].data_ptr(),
                     dnext.data_ptr(), dnext.stride(0), N, h, w, u["cin"], c, self.dt, s)
            self._notify(flat, [u["mod"]])
            dact = dnext
        # bottleneck: input is pool[4]
        self._block_bwd("bottleneck", flat, 4, dact, None, g[("dpool", 4)])
        # encoder, level 4 down to 1: gradient = skip half of dcat + scatter of the pooled gradient
        for lvl in (4, 3, 2, 1):
            l = lvl - 1
            c = [64, 128, 256, 512][l]
            name = ENCODER[l]
            dskip = g[("dcat", lvl)][:, c:]
            dx_out = g[("dpool", lvl - 1)] if lvl > 1 else None
            self._block_bwd(name, flat, l, dskip, g[("dpool", lvl)], dx_out)
        return flat


def _release_engine(ref, generation):
    eng = ref()
    if eng is not None and eng.busy and eng._busy_gen == generation:
        eng.busy = False


class _UNetFunction(torch.autograd.Function):
    """Autograd boundary: one node for the whole network (forward kernels / backward kernels)."""

    @staticmethod
    def forward(ctx, x, engine, frozen, *params):
        probs = engine.forward(x, train=True, frozen=frozen)
        ctx.engine = engine
        ctx.generation = engine.generation
        engine.busy, engine._busy_gen = True, engine.generation
        # a forward that is never back-propagated releases its engine when its graph is freed
        import weakref
        weakref.finalize(ctx, _release_engine, weakref.ref(engine), engine.generation)
        # x (read by the first layer's weight gradient) and probs (read by the head backward) go through autograd's
        # saved-tensor machinery so that an in-place edit of either between forward and backward is detected
        ctx.save_for_backward(x, probs)
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        eng = ctx.engine
        if eng.generation != ctx.generation:
            raise _lib.UnetdcError(
                "backward through a U-Net forward whose saved activations were overwritten by a later forward of "
                "the same module (forward #%d, buffers now hold #%d): the HIP path keeps ONE set of activation "
                "buffers per module, so run backward before the next forward (train or eval) of that module"
                % (ctx.generation, eng.generation))
        x, probs = ctx.saved_tensors
        flat = eng.backward(dprobs, probs, need_dx=ctx.needs_input_grad[0])
        finish = eng.model.grad_sync_finish
        if finish is not None:           # data parallel: wait (stream-side) for the bucket all-reduces
            finish()
        grads = []
        for p, o in zip(eng.params, eng.poffs):
            grads.append(flat[o:o + p.numel()].view_as(p) if p.requires_grad else None)
        dx = eng.input_grad() if ctx.needs_input_grad[0] else None
        # the gradients are enqueued: the next forward may take the buffers (a SECOND backward through a retained graph stays
        # possible as long as no forward has used them in between -- the generation check above says so otherwise)
        _release_engine(lambda: eng, ctx.generation)
        return (dx, None, None, *grads)
