set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "valid_rectangles or conv3x3_fwd_dgrad_wgrad" > gpurun_out/rect2_ops.log 2>&1 || { tail -30 gpurun_out/rect2_ops.log; exit 1; }
tail -2 gpurun_out/rect2_ops.log
bash tools/ab_bench.sh "wgrad 8 32 32 512 1024 16 bf16" 2 2>/dev/null
bash tools/ab_bench.sh "wgrad 8 32 32 1024 1024 16 bf16" 1 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/rect2_b.json; python3 tools/show_bench.py gpurun_out/rect2_b.json
