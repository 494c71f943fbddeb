#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + HBM PMC passes (separate runs, as the microarch guide
# prescribes) over the GLOBAL-BA part of bench.py only, with the driver's own --steps / --warmup.  Outputs under
# gpurun_out/prof_$1; scripts/collect_profiles.py turns them into the committed summaries under profiles/.
TAG=${1:-r02}
STEPS=${2:-20}
WARMUP=${3:-5}
EXTRA=${4:-}      # e.g. "--workload gba_c5" (collect_profiles.py then writes profiles/pmc_gba_c5.json instead of pmc_latest.json)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --gba-only --steps $STEPS --warmup $WARMUP $EXTRA"
echo "$CMD" > $OUT/command.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
grep -h '^{"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line_profiled_run.json
# the raw per-dispatch CSVs are large: keep compact per-(kernel, grid) aggregates next to them for the merge back
python scripts/collect_profiles.py $TAG --on-box
