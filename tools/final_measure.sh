#!/bin/bash
# Round-end measurement set (GPU box, repo root):  tools/final_measure.sh <tag>     e.g. r02_final
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
tag=${1:-r05_final}
export TMPDIR=/tmp
set -e
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err                      # default run (with cpu_baseline)
python3 tools/show_bench.py gpurun_out/${tag}_bench.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --per-layer > gpurun_out/${tag}_bench_nocpu.json 2> gpurun_out/${tag}_per_layer.txt
bash tools/rocprof_stats.sh ${tag}_prof > gpurun_out/${tag}_stats_summary.txt 2>&1
head -3 gpurun_out/${tag}_stats_summary.txt
bash tools/pmc_traffic.sh ${tag} > gpurun_out/${tag}_pmc_log.txt 2>&1 || tail -5 gpurun_out/${tag}_pmc_log.txt
python3 bench.py --mode infer --dtype f32 --steps 20 --no-secondary --warmup 5 > gpurun_out/${tag}_bench_infer_f32.json 2>/dev/null
python3 bench.py --mode infer --dtype bf16 --steps 20 --no-secondary --warmup 5 > gpurun_out/${tag}_bench_infer_bf16.json 2>/dev/null
python3 bench.py --size 1024 --batch 4 --steps 10 --no-secondary --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_1024.json 2>/dev/null
python3 bench.py --arch unet --steps 20 --no-secondary --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_unet.json 2>/dev/null
python3 bench.py --in-channels 3 --steps 20 --no-secondary --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_rgb.json 2>/dev/null
python3 bench.py --dtype f32 --steps 5 --no-secondary --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_train_f32.json 2>/dev/null
UNETDC_DP_FORCE=1 python3 bench.py --steps 20 --no-secondary --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_rccl1.json 2>/dev/null
python3 bench.py --mode quantify --steps 10 --warmup 3 > gpurun_out/${tag}_bench_quantify.json 2>/dev/null
ZERO=1 python3 tools/op_bench.py fwd 8 64 64 512 512 1 bf16 30 > gpurun_out/${tag}_zero_vs_random.txt 2>/dev/null
python3 tools/op_bench.py fwd 8 64 64 512 512 1 bf16 30 >> gpurun_out/${tag}_zero_vs_random.txt 2>/dev/null
ZERO=1 python3 tools/op_bench.py fwd 8 512 512 64 64 1 bf16 30 >> gpurun_out/${tag}_zero_vs_random.txt 2>/dev/null
python3 tools/op_bench.py fwd 8 512 512 64 64 1 bf16 30 >> gpurun_out/${tag}_zero_vs_random.txt 2>/dev/null
for f in gpurun_out/${tag}_bench_*.json; do python3 tools/show_bench.py $f; done
cat gpurun_out/${tag}_zero_vs_random.txt
