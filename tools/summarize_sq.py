#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_sq.sh into <dir>/summary.json (and stdout).

    python tools/summarize_sq.py gpurun_out/pmc_<tag> <tag> "<op_bench args>"

Per kernel (MFMA kernels only): average counter value per dispatch, and
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
      (matrix-pipe busy cycles summed over the chip's 1024 SIMDs -- 16 per v_mfma_f32_16x16x32_bf16, 32 per
       32x32x16, MI355X_MICROARCH.md cycle constants -- over the kernel-active cycles of one XCD x 1024).
The raw rocprofv3 output stays under gpurun_out/ (scratch); copy summary.json to profiles/ to keep it."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"tag": tag, "op_bench_args": args, "kernels": {}}
    for k, cs in agg.items():
        kl = k.lower()
        if not any(s in kl for s in ("igemm", "wgrad", "conv")):
            continue
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        d = {"dispatches_sampled": max(len(v) for v in cs.values()), "counters": avg}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("GRBM_GUI_ACTIVE"):
            # matrix-pipe busy cycles summed over the 1024 SIMDs / (kernel-active cycles x 1024 SIMDs);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
            d["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("SQ_BUSY_CU_CYCLES"):
            d["mfma_busy_over_busy_cu_cycles"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / avg["SQ_BUSY_CU_CYCLES"]
        if "SQ_WAVE_CYCLES" in avg and avg["SQ_WAVE_CYCLES"]:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in avg:
                    d[c.lower() + "_frac_of_wave_cycles"] = avg[c] / avg["SQ_WAVE_CYCLES"]
        res["kernels"][k] = d
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
    for k, d in res["kernels"].items():
        print(k[:100])
        for key, v in d.items():
            if key != "counters":
                print(f"   {key:40s} {v}")
        for c, v in sorted(d["counters"].items()):
            print(f"   {c:40s} {v:16.0f}")


if __name__ == "__main__":
    main()
