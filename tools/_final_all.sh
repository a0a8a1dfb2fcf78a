set -e
cd $GRAFT_REPO_ROOT
( time timeout -k 10 1100 python -m pytest tests -x -q -m gpu ) > gpurun_out/r02_final_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r02_final_gpu_tests.log; exit 1; }
tail -6 gpurun_out/r02_final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_final_smoke.log 2>&1 || { tail -20 gpurun_out/r02_final_smoke.log; exit 1; }
tail -2 gpurun_out/r02_final_smoke.log
