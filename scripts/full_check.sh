#!/bin/bash
# GPU box: whole GPU test suite, the default bench line, and (unless NOPROF=1) the GBA-only profile passes.  usage: full_check.sh <tag>
TAG=${1:-r02x}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/${TAG}_pytest.log 2>&1
( time timeout 600 python bench.py ) > gpurun_out/${TAG}_bench.log 2>&1
grep -h '^{"metric"' gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench_line.json
if [ -z "$NOPROF" ]; then bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; fi
tail -5 gpurun_out/${TAG}_pytest.log; tail -c 600 gpurun_out/${TAG}_bench_line.json
