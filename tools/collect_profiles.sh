#!/bin/bash
# Copy the round-end measurement set from gpurun_out/ (scratch) into profiles/ (tracked):
#   tools/collect_profiles.sh <round tag>        (in the container, after gpurun has merged the box's gpurun_out/)
r=${1:-r04}
set -e
for f in bench bench_nocpu bench_1024 bench_infer_bf16 bench_infer_f32 bench_quantify bench_rccl1 bench_rgb bench_train_f32 bench_unet; do
  cp gpurun_out/${r}_final_$f.json profiles/${r}_final_$f.json
done
cp gpurun_out/${r}_final_prof_kernel_stats.csv profiles/${r}_final_kernel_stats.csv
cp gpurun_out/${r}_final_stats_summary.txt gpurun_out/${r}_final_per_layer.txt gpurun_out/${r}_final_zero_vs_random.txt profiles/
cp gpurun_out/${r}_final_pmc_traffic.json profiles/${r}_pmc_traffic.json
cp gpurun_out/${r}_quantify_kernel_stats_after.csv profiles/
cp gpurun_out/${r}_rccl1_trace.txt profiles/${r}_rccl1_trace.txt
python3 - "$r" <<'PY'
import glob, json, os, sys
r = sys.argv[1]
out = {}
for d in sorted(glob.glob(f"gpurun_out/pmc_*_{r}")):
    f = os.path.join(d, "summary.json")
    if os.path.exists(f):
        out[os.path.basename(d)[4:]] = json.load(open(f))
json.dump(out, open(f"profiles/{r}_sq_counters.json", "w"), indent=1)
PY

