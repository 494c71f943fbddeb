#!/bin/bash
# development probe (GPU box): the round-5 and the round-6 form of the row Schur kernel on the SAME box (CCM_BA_ROW_OLD=1: staging by the first waves, every pass begins with
# its own requests, diagonal sums behind the second barrier) — per-class kernel times of five LM iterations of WORKLOAD (default gba_c4) and the chi2 after them to the last bit
cd $GRAFT_REPO_ROOT
W=${WORKLOAD:-gba_c4}
for rep in 1 2; do
for v in 1 0; do
  echo "== CCM_BA_ROW_OLD=$v"
  CCM_BA_ROW_OLD=$v python scripts/gpu_probe_ba.py $W 5 2>&1 | grep -E "run s|BA_SCHUR" | sed -e 's/ms_setup.*//'
done
done
