#!/bin/bash
# SQ counter set for the main MFMA kernels (GPU box, repo root): tools/sq_set.sh <suffix>
sfx=${1:-r03}
bash tools/pmc_sq.sh lat64_$sfx fwd 8 512 512 64 64 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh lat128_$sfx fwd 8 256 256 128 128 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh lat512_$sfx fwd 8 64 64 512 512 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh wsplit128_$sfx wgrad 8 256 256 128 128 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh wsplit512_$sfx wgrad 8 64 64 512 512 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh wrect_$sfx wgrad 8 32 32 1024 1024 16 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh convt_$sfx convt_fwd 8 32 32 1024 512 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh bott_$sfx fwd 8 32 32 1024 1024 16 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh wide512_$sfx fwd 8 64 64 512 512 1 bf16 > /dev/null 2>&1
ls gpurun_out | grep "pmc_.*_$sfx"
