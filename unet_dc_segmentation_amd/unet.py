"""Drop-in ``nn.Module`` surface for the reference's U-Net / U-Net-DC.

Boundary contract:
  * ``UNetDC(in_channels=3, out_channels=1)``  <-> /root/reference/models/model_2.py:5-32
  * ``UNet(in_channels=3, out_channels=1)``    <-> /root/reference/models/model.py:7-23
  * identical 136-key ``state_dict`` (``enc1.0.weight`` ... ``out_conv.bias``), identical default
    initialisation under a given ``torch.manual_seed`` (same constructors, same order), fp32
    ``nn.Parameter``s in PyTorch layout so ``torch.optim.Adam(model.parameters())`` works unchanged
    (train_DC_focal.py:224).
  * ``forward(x[N,C,H,W] fp32) -> probabilities[N,out,H,W] fp32`` (model_2.py:56-80), autograd
    capable, ``train()``/``eval()`` switch BatchNorm between batch and running statistics.

On a HIP device every arithmetic op runs in the hand-written gfx950 kernels of
``csrc/`` through :mod:`unet_dc_segmentation_amd.engine`; there is no PyTorch fallback there -- a
missing ``libunetdc_hip.so`` raises.  On CPU tensors (the reference's ``DEVICE = "cpu"`` case,
BASELINE config 0) the module runs the ordinary ATen CPU ops, which is what the reference does
on a host without a GPU.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

ENCODER = ("enc1", "enc2", "enc3", "enc4")
BLOCK_ORDER = ("enc1", "enc2", "enc3", "enc4", "bottleneck", "dec4", "dec3", "dec2", "dec1")
WIDTHS = {"enc1": 64, "enc2": 128, "enc3": 256, "enc4": 512, "bottleneck": 1024,
          "dec4": 512, "dec3": 256, "dec2": 128, "dec1": 64}


def _stage_pair(cin, cout, d):
    """Two (dilated 3x3 conv -> BatchNorm -> ReLU) stages with Sequential indices 0..5, so the
    state-dict keys are ``<block>.{0,1,3,4}.*`` exactly as in the reference."""
    mods = []
    for c in (cin, cout):
        mods += [nn.Conv2d(c, cout, kernel_size=3, padding=d, dilation=d),
                 nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
    return nn.Sequential(*mods)


class _UNetFamily(nn.Module):
    """Shared topology; subclasses only choose the per-block dilation."""

    DILATIONS: dict = {}

    def __init__(self, in_channels=3, out_channels=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        d = self.DILATIONS
        prev = in_channels
        for name in ENCODER + ("bottleneck",):           # registration order == reference order
            setattr(self, name, _stage_pair(prev, WIDTHS[name], d[name]))
            prev = WIDTHS[name]
        for lvl in (4, 3, 2, 1):
            w = WIDTHS[f"dec{lvl}"]
            setattr(self, f"upconv{lvl}", nn.ConvTranspose2d(2 * w, w, kernel_size=2, stride=2))
            setattr(self, f"dec{lvl}", _stage_pair(2 * w, w, d[f"dec{lvl}"]))
        self.out_conv = nn.Conv2d(64, out_channels, kernel_size=1)
        # compute configuration of the HIP path (not part of the reference surface):
        #   "f32"  -- fp32 storage, exact-fp32 MFMA (parity configuration)
        #   "bf16" -- bf16 activations/weights, fp32 accumulate (throughput configuration)
        self.compute_dtype = "f32"
        self._engines = {}               # (device, input shape) -> [engine.UNetEngine, ...], most recently used last
        self._weights = None             # engine.PackedWeights: compute-type weight images, shared by the engines
        self.grad_ready_hook = None      # set by the data-parallel wrapper (dp.py)
        self.grad_sync_finish = None

    # ------------------------------------------------------------------ configuration
    def set_compute_dtype(self, name):
        if name not in ("f32", "bf16"):
            raise ValueError(f"compute dtype must be 'f32' or 'bf16', got {name!r}")
        self.compute_dtype = name
        self._engines, self._weights = {}, None
        return self

    MAX_ENGINES = 4      # train_DC_focal.py: full batch, ragged last training batch, ragged last validation batch, ragged test batch

    def _engine_for(self, x):
        """One engine (activation buffers, schedule) per input shape, a few kept alive: the ragged last training and validation
        batches of train_DC_focal.py (its loaders have no drop_last, like the reference's) must not free and re-allocate the
        full-batch buffers every epoch.  The packed weight images are shared (engine.PackedWeights).

        An engine whose activations belong to a LIVE autograd graph (a forward under autograd whose backward has not run yet)
        is busy: another forward of the same shape -- a second micro-batch before the first backward, a validation forward
        between forward and backward -- gets its own engine instead of overwriting those activations, as the reference's plain
        autograd module allows (models/model_2.py:56-80).  Sized for 288 GB: ~12 GB per live bs-8 512 x 512 bf16 forward."""
        from . import engine                           # raises if libunetdc_hip.so is missing
        key = (x.device, tuple(x.shape))
        pool = self._engines.pop(key, [])
        pool = [e for e in pool if e.matches(x)]
        eng = next((e for e in pool if not e.busy), None)
        if eng is None:
            eng = engine.UNetEngine(self, x, weights=self._weights)
            self._weights = eng.weights
            pool.append(eng)
            # least recently used shapes go first; busy engines are never dropped here (their graph still needs them)
            while sum(len(v) for v in self._engines.values()) + len(pool) > self.MAX_ENGINES:
                victim = next(((k, e) for k, v in self._engines.items() for e in v if not e.busy), None)
                if victim is None:
                    extra = next((e for e in pool if e is not eng and not e.busy), None)
                    if extra is None:
                        self._warn_live_graphs(sum(len(v) for v in self._engines.values()) + len(pool), eng)
                        break
                    pool.remove(extra)
                    continue
                self._engines[victim[0]].remove(victim[1])
                if not self._engines[victim[0]]:
                    del self._engines[victim[0]]
        else:
            pool.remove(eng)
            pool.append(eng)                           # most recently used last
        self._engines[key] = pool
        return eng

    def _warn_live_graphs(self, live, eng):
        """Every engine is held by a live autograd graph and the pool is past MAX_ENGINES: say so once -- a caller that keeps
        outputs with gradients enabled (predictions collected in a list without torch.no_grad()) otherwise runs the device out
        of memory with an error that names nothing."""
        if getattr(self, "_warned_live", False):
            return
        self._warned_live = True
        import warnings
        per = sum(t.numel() * t.element_size() for t in
                  [st.y for st in eng.stages.values()] + list(eng.a0.values()) + list(eng.a3.values())
                  + list(eng.cat.values()) + list(eng.pool.values())) / 2 ** 30
        warnings.warn(f"{type(self).__name__}: {live} forwards of this module are alive at once (each keeps ~{per:.1f} GiB of "
                      f"activations for its backward; only {self.MAX_ENGINES} sets are kept for re-use).  If no backward is "
                      "meant to follow, run the forward under torch.no_grad(); otherwise call backward() (or drop the "
                      "outputs) before starting further forwards.", RuntimeWarning, stacklevel=4)

    def __getstate__(self):
        """copy.deepcopy / pickling (torch.save(model)) carry the module, not the device-side engines: activation buffers,
        packed weight images and the data-parallel hooks are rebuilt on the first forward of the copy."""
        state = dict(self.__dict__)
        state["_engines"], state["_weights"] = {}, None
        state["grad_ready_hook"], state["grad_sync_finish"] = None, None
        return state

    @property
    def _engine(self):
        """The most recently used engine (None before the first HIP forward)."""
        return next(reversed(self._engines.values()))[-1] if self._engines else None

    def dilation_of(self, block):
        return self.DILATIONS[block]

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if x.is_cuda:
            return self._engine_for(x).run(x)
        return self._forward_aten_cpu(x)

    def _forward_aten_cpu(self, x):
        skips = []
        h = x
        for name in ENCODER:
            h = getattr(self, name)(h)
            skips.append(h)
            h = F.max_pool2d(h, 2)
        h = self.bottleneck(h)
        for lvl in (4, 3, 2, 1):
            up = getattr(self, f"upconv{lvl}")(h)
            h = getattr(self, f"dec{lvl}")(torch.cat([up, skips[lvl - 1]], dim=1))
        return torch.sigmoid(self.out_conv(h))


class UNetDC(_UNetFamily):
    """Dilated U-Net: encoder dilations 1/2/4/8, bottleneck 16, decoder 1 (model_2.py:10-30)."""
    DILATIONS = {"enc1": 1, "enc2": 2, "enc3": 4, "enc4": 8, "bottleneck": 16,
                 "dec4": 1, "dec3": 1, "dec2": 1, "dec1": 1}


class UNet(_UNetFamily):
    """Plain U-Net: every dilation 1 (models/model.py:25-33)."""
    DILATIONS = {b: 1 for b in BLOCK_ORDER}
