"""Build ``libunetdc_hip.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["abi.hip", "igemm_conv.hip", "igemm_dma.hip", "igemm_dma16.hip", "igemm_halo.hip", "igemm_lattice.hip", "wgrad.hip", "wgrad_dma.hip", "wgrad_fused.hip", "wgrad_rect.hip", "first_conv.hip", "first_conv_mfma.hip", "elementwise.hip", "loss.hip", "optim.hip", "ccl.hip", "preprocess.hip"]
HEADERS = ["common.h", "kernels.h", "lds_dma.h", "wgrad_frag.h", "igemm_epilogue.h", "igemm_epilogue16.h", os.path.join("..", "..", "include", "unetdc_hip.h")]
OUT = os.path.join(HERE, "libunetdc_hip.so")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the gfx950 kernels)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    # -fno-slp-vectorize: keeps the channel-pair arithmetic scalar (no v_pk_*_f32).  The packed build of the
    # fused BatchNorm-backward epilogue was not run-to-run deterministic on MI355X (csrc/igemm_epilogue.h,
    # BUILD NOTE); the scalar build is, and measures the same speed.
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-fno-slp-vectorize", *[os.path.join(CSRC, f) for f in SOURCES], "-o", OUT]
    if verbose:
        print("[unetdc build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libunetdc_hip.so")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
