"""Build ``libunetdc_hip.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["abi.hip", "igemm_conv.hip", "igemm_dma.hip", "igemm_dma16.hip", "igemm_halo.hip", "igemm_lattice.hip", "wgrad.hip", "wgrad_dma.hip", "wgrad_fused.hip", "wgrad_rect.hip", "convt_wgrad.hip", "first_conv.hip", "first_conv_mfma.hip", "elementwise.hip", "loss.hip", "optim.hip", "ccl.hip", "preprocess.hip"]
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


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]
OBJ_DIR = os.path.join(CSRC, "_obj")


def _includes(path, seen=None):
    """Transitive closure of the quoted #include files of one translation unit (for the per-object cache key)."""
    import re
    seen = set() if seen is None else seen
    for m in re.finditer(r'^\s*#include\s+"([^"]+)"', open(path).read(), re.M):
        q = os.path.normpath(os.path.join(os.path.dirname(path), m.group(1)))
        if q not in seen and os.path.exists(q):
            seen.add(q)
            _includes(q, seen)
    return seen


def _object_key(src):
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for q in [src] + sorted(_includes(src)):
        h.update(open(q, "rb").read())
    return h.hexdigest()[:16]


def build(force=False, verbose=True):
    """One hipcc -c per translation unit (in parallel, objects cached under csrc/_obj by a hash of the source, its
    headers and the flags), then one link."""
    if not force and not needs_build():
        return OUT
    # -fno-slp-vectorize: keeps the channel-pair arithmetic scalar (no v_pk_*_f32).  The packed build of the
    # fused BatchNorm-backward epilogue was not run-to-run deterministic on MI355X (csrc/igemm_epilogue.h,
    # BUILD NOTE); the scalar build is, and measures the same speed.
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()

    def one(f):
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ_DIR, f"{os.path.splitext(f)[0]}.{_object_key(src)}.o")
        if force or not os.path.exists(obj):
            for old in os.listdir(OBJ_DIR):
                if old.startswith(os.path.splitext(f)[0] + "."):
                    os.unlink(os.path.join(OBJ_DIR, old))
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print("[unetdc build]", " ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed on {f}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libunetdc_hip.so")
    return OUT


def _objdump():
    for cand in (shutil.which("llvm-objdump"), "/opt/rocm/lib/llvm/bin/llvm-objdump"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("llvm-objdump not found")


def device_code_objects(so_path=OUT):
    """The gfx950 ELF images inside the library: one clang offload bundle per translation unit in .hip_fatbin
    (header: magic[24], count u64, then per entry offset u64, size u64, id-length u64, id)."""
    import struct
    data = open(so_path, "rb").read()
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        n, = struct.unpack_from("<Q", data, i + 24)
        q = i + 32
        for _ in range(n):
            off, size, idl = struct.unpack_from("<QQQ", data, q)
            q += 24
            tid = data[q:q + idl].decode()
            q += idl
            if "gfx950" in tid and size:
                out.append(data[i + off:i + off + size])
        pos = i + 24


def packed_f32_report(so_path=OUT):
    """{kernel symbol: number of packed-fp32 VALU instructions (v_pk_add/mul/fma_f32)} over every kernel of the library.
    The convolution kernels must have none: the packed build of the fused BatchNorm-backward epilogue was not run-to-run
    deterministic (build() above passes -fno-slp-vectorize; tests/test_entry_points_cpu.py keeps a toolchain or flag
    change from silently bringing the packed code back)."""
    import re
    import tempfile
    rep = {}
    for k, elf in enumerate(device_code_objects(so_path)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            r = subprocess.run([_objdump(), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("llvm-objdump failed: " + r.stderr[:300])
        cur = None
        for line in r.stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                rep.setdefault(cur, 0)
            elif cur and re.search(r"\bv_pk_(add|mul|fma)_f32\b", line):
                rep[cur] += 1
    return rep


if __name__ == "__main__":
    build(force="--force" in sys.argv)
