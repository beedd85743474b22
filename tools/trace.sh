#!/bin/bash
# usage (GPU box, repo root): tools/trace.sh <tag> [bench args] -- kernel trace + stats only
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 20 --warmup 5 --cpu-steps 0 "$@" > $out/trace.log 2>&1 || exit 1
python3 $root/tools/stats.py $out/trace 25 45 > $out/summary.txt
