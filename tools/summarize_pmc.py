#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/<tag>_pmc_traffic.json.

    python tools/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

FETCH_SIZE and WRITE_SIZE are collected in two SEPARATE passes (they do not fit one pass on gfx950:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  Per-launch averages in KiB; corrected bytes apply the
guide's gfx950 rule for 16-byte-per-lane streaming reads (FETCH_SIZE x 2).  The file is stamped with the sha256 of the
kernel sources (bench.kernel_source_hash): bench.py reports `roofline.traffic` from it only while the stamp matches
the sources it was built from.  `by_grid` splits every kernel by launch grid (= by layer shape)."""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def demangle(name):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if name.startswith("_Z") and tool:
        return subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()
    return name


def load(path, counter):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]][int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    import bench
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) over `bench.py --steps 3 "
           "--warmup 2 --no-cpu-baseline`; per-launch averages in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): "
           "FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming reads (global_load and buffer_load..lds "
           "alike) at 64 B, so bytes fetched = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.  Both count memory-side requests "
           "past the XCD L2s (Infinity-Cache hits included): an upper bound on HBM traffic.  by_grid: the same per launch "
           "grid (workgroups), i.e. per layer shape.",
           "kernel_source_sha16": bench.kernel_source_hash(), "kernels": {}}

    def avg(v):
        return sum(v) / len(v) if v else 0.0
    for name in sorted(set(f) | set(w)):
        fa = [x for g in f.get(name, {}).values() for x in g]
        wa = [x for g in w.get(name, {}).values() for x in g]
        entry = {"launches_sampled": len(fa), "FETCH_SIZE_KiB": avg(fa), "WRITE_SIZE_KiB": avg(wa),
                 "bytes_corrected": (2 * avg(fa) + avg(wa)) * 1024, "by_grid": {}}
        for grid in sorted(set(f.get(name, {})) | set(w.get(name, {}))):
            fg, wg = f.get(name, {}).get(grid, []), w.get(name, {}).get(grid, [])
            entry["by_grid"][str(grid)] = {"launches_sampled": len(fg), "bytes_corrected": (2 * avg(fg) + avg(wg)) * 1024}
        res["kernels"][demangle(name)] = entry
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["bytes_corrected"] * kv[1]["launches_sampled"])[:8]:
        print(f"{k[:90]:90s} {v['bytes_corrected'] / 1e6:9.1f} MB/launch x{v['launches_sampled']}")


if __name__ == "__main__":
    main()
