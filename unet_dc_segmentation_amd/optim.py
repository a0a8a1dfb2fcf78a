"""Fused Adam for the HIP path: ONE kernel per step updates every fp32 master parameter and rewrites the packed
compute-type weight images the conv kernels read (csrc/optim.hip, C ABI ``unetdc_adam_step``).

Reference: ``optimizer = optim.Adam(model.parameters(), lr=0.001)`` ... ``optimizer.step()``
(/root/reference/train_DC_focal.py:224,255; train.py:125,157).  Same update rule and defaults as
``torch.optim.Adam`` (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad); ``state_dict()`` carries the same
per-parameter entries (``step``, ``exp_avg``, ``exp_avg_sq``), so a checkpointed optimizer state moves between the two.

What is saved against torch.optim.Adam(fused=True) + the engine's re-pack launch: the 124 MB of parameters are read and
written once per step instead of three times, and the gradients are read straight out of the flat buffer the backward
kernels wrote (``grad_scale`` folds the 1/G of a SUM all-reduce into that read).  On CPU tensors (or any parameter set
that does not belong to one HIP module) the class falls back to the plain per-tensor formulas in PyTorch -- that is the
reference's own code path on a host without a GPU, not a fallback of the HIP path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_DESC = np.dtype([("p", "<u8"), ("m", "<u8"), ("v", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("g_off", "<i8"),
                  ("begin", "<i8"), ("numel", "<i8"), ("a", "<i4"), ("b", "<i4"), ("kind", "<i4"), ("pad", "<i4")])
assert _DESC.itemsize == 80


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        if not isinstance(model, torch.nn.Module):
            raise TypeError("FusedAdam takes the U-Net module (it needs the module's packed weight images), "
                            "not a parameter list")
        self.model = model
        # the full set of torch.optim.Adam defaults: a state_dict() of this class loads into torch.optim.Adam (whose
        # step() reads every one of these keys from the group) and the other way round
        super().__init__(list(model.parameters()),
                         dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                              capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))
        self.grad_scale = float(grad_scale)
        self._step = 0
        self._table = None
        self._table_key = None

    # ------------------------------------------------------------------ state
    def _init_state(self):
        params = self.param_groups[0]["params"]
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self._m = torch.zeros(n, device=dev, dtype=torch.float32)
        self._v = torch.zeros(n, device=dev, dtype=torch.float32)
        self._offs, o = [], 0
        # ONE step counter object shared by every parameter's state entry (torch.optim.Adam keeps one per parameter, all
        # equal): 82 separate `step += 1` on 0-dim CPU tensors cost ~0.8 ms of host time per training step
        self._step_t = torch.tensor(float(self._step))
        for p in params:
            self._offs.append(o)
            st = self.state[p]
            st["step"] = self._step_t
            st["exp_avg"] = self._m[o:o + p.numel()].view_as(p)
            st["exp_avg_sq"] = self._v[o:o + p.numel()].view_as(p)
            o += p.numel()
        self._n = n

    def state_dict(self):
        """torch.optim.Adam's layout: every parameter's entry carries its OWN ``step`` tensor (the shared counter object is
        an internal economy; handing it out would make a non-fused torch Adam advance it once per parameter)."""
        sd = super().state_dict()
        # torch.optim.Optimizer.state_dict() hands out the LIVE per-parameter dicts: build copies, never assign into them
        sd["state"] = {k: (dict(st, step=torch.tensor(float(self._step))) if "step" in st else st)
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        for g in state_dict.get("param_groups", ()):
            if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
                raise ValueError("FusedAdam implements plain Adam only (weight_decay=0, amsgrad=False, maximize=False)")
        super().load_state_dict(state_dict)
        params = self.param_groups[0]["params"]
        if params and "exp_avg" in self.state.get(params[0], {}):
            loaded = [(self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in params]
            self._step = int(float(self.state[params[0]]["step"]))
            self._init_state()                     # flat buffers; copy the loaded moments into them
            for p, (ea, es) in zip(params, loaded):
                self.state[p]["exp_avg"].copy_(ea)
                self.state[p]["exp_avg_sq"].copy_(es)
            self._table = None

    # ------------------------------------------------------------------ descriptor table
    def _build_table(self, weights):
        params = self.param_groups[0]["params"]
        packed = {}
        if weights is not None:
            for w, wf, wd, a, b, kind in weights.entries():
                packed[id(w)] = (wf, wd, a, b, kind)
        tab = np.zeros(len(params), dtype=_DESC)
        begin = 0
        for i, (p, off) in enumerate(zip(params, self._offs)):
            m_ptr = self._m.data_ptr() + 4 * off
            v_ptr = self._v.data_ptr() + 4 * off
            if id(p) in packed:
                wf, wd, a, b, kind = packed[id(p)]
                tab[i] = (p.data_ptr(), m_ptr, v_ptr, wf.data_ptr(), wd.data_ptr(), off, begin, p.numel(), a, b, kind, 0)
                begin += (a // 32) * (b // 32)
            else:
                tab[i] = (p.data_ptr(), m_ptr, v_ptr, 0, 0, off, begin, p.numel(), 0, 0, 2, 0)
                begin += (p.numel() + 4095) // 4096
        self._table = torch.from_numpy(tab.view(np.uint8).copy()).to(params[0].device)
        self._blocks = begin

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        params = group["params"]
        if any(p.grad is None for p in params):
            raise RuntimeError("FusedAdam.step(): every parameter needs a gradient (one backward through the module)")
        if not hasattr(self, "_m"):
            self._init_state()
        self._step += 1
        lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
        if not params[0].is_cuda:
            self._step_torch(params, lr, b1, b2, eps)
            return loss
        # gradients: normally the views autograd got from the HIP backward, i.e. ONE flat buffer in parameters() order
        base = params[0].grad.data_ptr()
        flat_ok = all(p.grad.is_contiguous() and p.grad.dtype == torch.float32 and
                      p.grad.data_ptr() == base + 4 * off for p, off in zip(params, self._offs))
        if flat_ok:
            flat_ptr, keep = base, None
        else:
            keep = torch.cat([p.grad.reshape(-1).float() for p in params])
            flat_ptr = keep.data_ptr()
        # the module's packed weight images (engine.PackedWeights: one set per module, shared by the engines of every input
        # shape); keyed by its never-reused serial, not id(): a rebuilt object may be handed a freed one's address
        weights = getattr(self.model, "_weights", None)
        key = (getattr(weights, "serial", None), tuple(p.data_ptr() for p in params))
        if self._table is None or self._table_key != key:
            self._build_table(weights)
            self._table_key = key
        dt = weights.dt if weights is not None else _lib.F32
        _lib.call("unetdc_adam_step", self._table.data_ptr(), len(params), self._blocks, flat_ptr, float(lr), float(b1),
                  float(b2), float(eps), self._step, self.grad_scale, dt, torch.cuda.current_stream().cuda_stream)
        if weights is not None:
            # both packed images were just rewritten from the parameters as they are NOW: the next forward skips its own
            # re-pack only while these version counters still stand (load_state_dict / copy_ / clamp_ after this step bump them)
            weights.fresh(tuple(w._version for w, *_ in weights.entries()))
        self._step_t += 1
        return loss

    def _step_torch(self, params, lr, b1, b2, eps):
        bc1, bc2 = 1.0 - b1 ** self._step, 1.0 - b2 ** self._step
        for p in params:
            st = self.state[p]
            g = p.grad * self.grad_scale if self.grad_scale != 1.0 else p.grad
            st["exp_avg"].lerp_(g, 1.0 - b1)
            st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (st["exp_avg_sq"].sqrt() / (bc2 ** 0.5)).add_(eps)
            p.addcdiv_(st["exp_avg"], denom, value=-lr / bc1)
        self._step_t += 1
