set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or conv3x3" > gpurun_out/rows_ops.log 2>&1 || { tail -40 gpurun_out/rows_ops.log; exit 1; }
tail -2 gpurun_out/rows_ops.log
bash tools/rocprof_stats.sh r02_rows > gpurun_out/rows_stats.txt 2>&1 || true
head -12 gpurun_out/rows_stats.txt
grep -E "finalize|colsum|reduce" gpurun_out/rows_stats.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --per-layer > gpurun_out/b_rows.json 2> gpurun_out/b_rows_layers.txt
python tools/show_bench.py gpurun_out/b_rows.json
tail -22 gpurun_out/b_rows_layers.txt
