python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r04f_ops.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/r04f_ops.log
bash tools/ab_lib.sh r04f_wgrad_reorder -- "wgrad 8 64 64 512 512 1 bf16" "wgrad 8 256 256 128 128 1 bf16" "wgrad 8 128 128 256 256 4 bf16" "wgrad_bnin 8 512 512 64 64 1 bf16" "convt_wgrad 8 64 64 512 256 1 bf16" "convt_wgrad 8 256 256 128 64 1 bf16"
bash tools/pmc_sq.sh wide512_r04 fwd 8 64 64 512 512 1 bf16 > /dev/null 2>&1; cat gpurun_out/pmc_wide512_r04/summary.json | head -60
