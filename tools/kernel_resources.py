#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in libunetdc_hip.so, from the code objects' metadata notes (no GPU needed).

    python tools/kernel_resources.py [substring ...]      e.g.  python tools/kernel_resources.py lattice convt
A non-zero scratch size means spills: in the kernels with hand-counted vmcnt waits a scratch reload drains the VM queue."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_dc_segmentation_amd import build as B  # noqa: E402


def kernels(so_path=B.OUT):
    readelf = os.path.join(os.path.dirname(B._objdump()), "llvm-readelf")
    out = []
    for elf in B.device_code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- ", txt):
            m = re.search(r"\.name:\s+(\S+)", blk)
            if not m or ".vgpr_count" not in blk:
                continue
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, blk) else -1  # noqa: E731
            out.append(dict(name=m.group(1), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                            scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                            spill_v=g("vgpr_spill_count"), spill_s=g("sgpr_spill_count")))
    return out


if __name__ == "__main__":
    pats = sys.argv[1:]
    dem = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in kernels()), capture_output=True, text=True).stdout.splitlines()
    for k, d in zip(kernels(), dem):
        if pats and not any(p in d for p in pats):
            continue
        print(f"{d[:110]:110s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} scratch {k['scratch']:4d} "
              f"spill v/s {k['spill_v']}/{k['spill_s']}")
