#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
MTG_DL_OCC2=0 bash tools/gpu_pmc_stalls.sh r03f_dl_occ1 --steps 30 --warmup 10 --dims dimlane --batch 125000 --sequence launches
MTG_DL_OCC2=1 bash tools/gpu_pmc_stalls.sh r03f_dl_occ2 --steps 30 --warmup 10 --dims dimlane --batch 125000 --sequence launches
bash tools/gpu_pmc_stalls.sh r03f_slab --steps 30 --warmup 10 --batch 125000 --sequence launches
