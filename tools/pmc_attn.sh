#!/bin/bash
# usage (GPU box): tools/pmc_attn.sh <tag> -- SQ counter passes for per-kernel pipe utilisation
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/p1 -- python3 $root/bench.py --steps 3 --warmup 2 --cpu-steps 0 "$@" > $out/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/p2 -- python3 $root/bench.py --steps 3 --warmup 2 --cpu-steps 0 "$@" > $out/p2.log 2>&1 || exit 1
ls $out/p1/*/ | head
