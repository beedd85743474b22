#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args]
# kernel trace + stats, then three PMC passes (SQ, FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 20 --warmup 5 --cpu-steps 0 "$@" > $out/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $root/bench.py --steps 4 --warmup 2 --cpu-steps 0 "$@" > $out/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 4 --warmup 2 --cpu-steps 0 "$@" > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 4 --warmup 2 --cpu-steps 0 "$@" > $out/pmc_write.log 2>&1 || exit 1
ls $out/*/*/ | head -30
