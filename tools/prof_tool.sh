#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_tool.sh <tag> <tool.py> [args]
# kernel trace + stats and one SQ counter pass of a micro-benchmark under tools/ (same recipe as tools/prof.sh)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
tool=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $tool "$@" > $out/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $tool "$@" > $out/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc_inst -- python3 $tool "$@" > $out/pmc_inst.log 2>&1 || true
cd $root && python3 tools/pmc.py gpurun_out/$tag 12 > $out/pmc_summary.txt 2>&1
cat $out/pmc_summary.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$out/trace/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("%-70s n=%s avg=%.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
