"""ctypes binding of ``libunetdc_hip.so`` (C ABI declared in ``include/unetdc_hip.h``).

The library is loaded AFTER ``import torch`` so that its ``libamdhip64.so.7`` dependency resolves
to the HIP runtime PyTorch-ROCm already has in the process (same SONAME): kernels are then
enqueued on PyTorch's own streams.  There is no fallback: if the shared object is missing or a
symbol is absent, importing/using the HIP path raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

import torch  # noqa: F401  (must be imported before the CDLL below -- see module docstring)

F32, BF16 = 0, 1
LIB_NAME = "libunetdc_hip.so"
LIB_PATH = os.environ.get("UNETDC_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

P, I, L, F, D = c_void_p, c_int, c_int64, c_float, c_double

# name -> (restype, argtypes); mirrors include/unetdc_hip.h one to one
SIGNATURES = {
    "unetdc_version": (I, []),
    "unetdc_last_error": (c_char_p, []),
    "unetdc_last_kernel": (c_char_p, []),
    "unetdc_pack_conv3x3": (I, [P, P, P, I, I, I, P]),
    "unetdc_pack_convT2x2": (I, [P, P, P, I, I, I, P]),
    "unetdc_pack_many": (I, [P, I, L, I, P]),
    "unetdc_adam_step": (I, [P, I, L, P, D, D, D, D, L, D, I, P]),
    "unetdc_conv3x3_stats_rows": (I, [L, I]),
    "unetdc_conv3x3_fwd": (I, [P, I, P, P, P, P, P, I, P, P, I, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_bnin_supported": (I, [I, I, I, I, I, I, I]),
    "unetdc_conv3x3_fwd_bnin": (I, [P, I, P, P, P, P, P, I, P, P, P, I, I, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_wgrad_bnin": (I, [P, I, P, P, P, I, P, P, L, I, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_dgrad": (I, [P, I, P, P, I, I, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_wgrad_workspace": (L, [I, I, I, I, I, I]),
    "unetdc_conv3x3_wgrad": (I, [P, I, P, I, P, P, L, I, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_first_stats_rows": (I, [L, I, I]),
    "unetdc_conv3x3_first_fwd": (I, [P, P, P, P, P, P, I, P, I, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_first_wgrad_workspace": (L, [I, I, I, I, I]),
    "unetdc_conv3x3_first_wgrad": (I, [P, P, I, P, P, L, I, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_first_dgrad": (I, [P, I, P, P, I, I, I, I, I, I, I, P]),
    "unetdc_convT2x2_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, I, P]),
    "unetdc_convT2x2_dgrad": (I, [P, I, P, P, I, I, I, I, I, I, I, P]),
    "unetdc_convT2x2_wgrad_workspace": (L, [I, I, I, I, I, I]),
    "unetdc_convT2x2_wgrad": (I, [P, I, P, I, P, P, L, I, I, I, I, I, I, P]),
    "unetdc_bn_finalize": (I, [P, I, L, P, P, F, F, P, P, P, P, P, P, I, P]),
    "unetdc_bn_eval_affine": (I, [P, P, P, P, P, F, P, P, I, P]),
    "unetdc_bn_relu_apply": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, P]),
    "unetdc_bn_relu_bwd_workspace": (L, [I, I, I, I, I, I]),
    "unetdc_bn_relu_bwd": (I, [P, I, P, I, P, I, P, P, P, P, P, P, I, P, P, P, P, L, P, I, I, I, I, I, I, P]),
    "unetdc_bn_frozen_affine": (I, [P, P, P, P, F, P, P, P, P, I, P]),
    "unetdc_bn_relu_bwd_head": (I, [P, P, P, P, I, P, P, P, P, P, P, I, P, P, P, P, L, P, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_first_wgrad_bn_supported": (I, [I, I, I, I, I, I, I]),
    "unetdc_bn_relu_bwd_coeffs": (I, [P, I, P, P, P, P, P, P, I, I, I, I, P]),
    "unetdc_conv3x3_first_wgrad_bn": (I, [P, P, I, P, I, P, P, P, P, P, P, P, L, I, I, I, I, I, I, I, P]),
    "unetdc_bn_relu_bwd_frozen": (I, [P, I, P, I, P, I, P, P, P, P, P, P, I, P, P, P, P, L, P, I, I, I, I, I, I, P]),
    "unetdc_conv3x3_dgrad_bnstats": (I, [P, I, P, P, I, P, I, P, P, P, P, P, L, P, I, I, I, I, I, I, I, P]),
    "unetdc_convT2x2_dgrad_bnstats": (I, [P, I, P, P, I, P, I, P, P, P, P, P, L, P, I, I, I, I, I, I, P]),
    "unetdc_head_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, P]),
    "unetdc_head_fwd_bn": (I, [P, I, P, P, P, P, P, I, I, I, I, I, I, P]),
    "unetdc_head_bwd_workspace": (L, [I, I, I, I, I, I]),
    "unetdc_head_bwd": (I, [P, P, P, I, P, P, I, P, P, P, L, I, I, I, I, I, I, P]),
    "unetdc_head_bwd_bnstats": (I, [P, P, P, I, P, P, I, P, P, P, L, P, I, P, P, P, P, P, L, P, I, I, I, I, I, I, P]),
    "unetdc_focal_dice_loss_workspace": (L, [I, L]),
    "unetdc_focal_dice_loss_fwd": (I, [P, P, P, P, P, L, I, L, F, F, F, F, P]),
    "unetdc_focal_dice_loss_bwd": (I, [P, P, P, P, P, I, L, F, F, F, P]),
    "unetdc_conv3x3_dgrad_colsum_workspace": (L, [I, I, I, I]),
    "unetdc_conv3x3_dgrad_colsum": (I, [P, I, P, P, I, P, I, I, P, L, I, I, I, I, I, I, I, P]),
    "unetdc_channel_sum_workspace": (L, [L, I]),
    "unetdc_channel_sum": (I, [P, I, P, P, L, L, I, I, P]),
    "unetdc_rolling_ball_workspace": (L, [I, I, I]),
    "unetdc_rolling_ball_u8": (I, [P, P, I, I, I, I, P, L, P]),
    "unetdc_resize_linear_u8_to_chw_f32": (I, [P, I, I, I, P, I, I, P, P, P, P, P]),
    "unetdc_mask_from_probs": (I, [P, I, I, F, P, I, I, P]),
    "unetdc_mask_from_probs_linear": (I, [P, I, I, F, P, I, I, P, P, P, P, P]),
    "unetdc_ccl_workspace": (L, [I, I]),
    "unetdc_ccl_stats": (I, [P, I, I, I, P, L, P, P, P, P, P, I, P]),
}

_lib = None


class UnetdcError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UnetdcError(
            f"{LIB_PATH} not found: the MI355X HIP library has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU/PyTorch "
            "fallback for tensors on a HIP device.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().unetdc_last_error().decode(errors="replace")
        raise UnetdcError(f"{what or 'unetdc'} failed (code {rc}): {msg}")


# ---- optional per-call timing (bench.py's instrumented passes, which run AFTER its timed region): HIP events recorded on
# the stream the kernels are launched on (PyTorch's current stream), bracketing selected C-ABI calls.  Outside
# start_timing() ... stop_timing() a call creates no event and takes the one-line path at the bottom of call().
_timing = None
events_created = 0          # total timing events ever created by call() (tests assert it stands still over a timed region)


def start_timing(names):
    global _timing
    _timing = {"names": set(names), "records": []}


def stop_timing():
    """Returns [(name, args, milliseconds)] for every bracketed call since start_timing()."""
    global _timing
    t, _timing = _timing, None
    if t is None:
        return []
    torch.cuda.synchronize()
    return [(n, a, e0.elapsed_time(e1)) for n, a, e0, e1 in t["records"]]


def call(name, *args):
    t = _timing
    if t is not None and name in t["names"]:
        global events_created
        events_created += 2
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args)
        e1.record()
        t["records"].append((name + "|" + load().unetdc_last_kernel().decode(), args, e0, e1))
        check(rc, name)
        return
    check(getattr(load(), name)(*args), name)
