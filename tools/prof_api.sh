#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_api.sh <tag> [config]
# rocprofv3 kernel trace + the three PMC passes of tools/prof.sh for the FORWARD / EVALUATION API path (tools/bench_api_forward.py:
# b4r_forward with materialised logits, the ranking kernels) -> gpurun_out/<tag>_<cfg>_api/summary/<tag>_{kernel_stats,pmc,build}_<cfg>_api.*
# (bench.py reads roofline_materialising.traffic from <tag>_pmc_<cfg>_api.txt and checks the build hash beside it)
tag=$1; cfg=${2:-ml1m}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}_${cfg}_api
rm -rf $out
mkdir -p $out/summary
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/bench_api_forward.py $cfg 20 > $out/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $root/tools/bench_api_forward.py $cfg 4 > $out/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/tools/bench_api_forward.py $cfg 4 > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/tools/bench_api_forward.py $cfg 4 > $out/pmc_write.log 2>&1 || exit 1
cd $root
python3 - > $out/summary/${tag}_build_${cfg}_api.json <<PY
import hashlib, json, os
lib = os.environ.get("B4R_LIB_PATH") or "bert4rec_amd/libb4r_hip.so"
print(json.dumps({"lib": lib, "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]}))
PY
cp $out/trace/*/*_kernel_stats.csv $out/summary/${tag}_kernel_stats_${cfg}_api.csv
python3 tools/pmc.py gpurun_out/${tag}_${cfg}_api 40 > $out/summary/${tag}_pmc_${cfg}_api.txt
cat $out/summary/${tag}_pmc_${cfg}_api.txt
