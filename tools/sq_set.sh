#!/bin/bash
# SQ counter set for the main MFMA kernels (GPU box, repo root): tools/sq_set.sh <suffix>
# Round 4: + the level-1 (64-channel) forms the round-3 verdict asked for, + the tap-fused ConvTranspose2d weight gradient.
sfx=${1:-r05}
bash tools/pmc_sq.sh lat64_$sfx fwd 8 512 512 64 64 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh lat64bnin_$sfx fwd_bnin 8 512 512 64 64 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh lat64bnbwd_$sfx dgrad_bnstats 8 512 512 64 64 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh wsplit64bnin_$sfx wgrad_bnin 8 512 512 64 64 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh lat128_$sfx fwd 8 256 256 128 128 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh lat128bnin_$sfx fwd_bnin 8 256 256 128 128 2 bf16 > /dev/null 2>&1     # round 5: wide form, normalise on load + activation write-back
bash tools/pmc_sq.sh lat512_$sfx fwd 8 64 64 512 512 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh latbnbwd256_$sfx dgrad_bnstats 8 128 128 256 256 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh wsplit128_$sfx wgrad 8 256 256 128 128 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh wsplit512_$sfx wgrad 8 64 64 512 512 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh wrect_$sfx wgrad 8 32 32 1024 1024 16 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh convt_$sfx convt_fwd 8 32 32 1024 512 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh convtwgrad_$sfx convt_wgrad 8 32 32 1024 512 1 bf16 > /dev/null 2>&1
bash tools/pmc_sq.sh bott_$sfx fwd 8 32 32 1024 1024 16 bf16 > /dev/null 2>&1
ls gpurun_out | grep "pmc_.*_$sfx"
