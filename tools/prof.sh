#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args, e.g. --config steam]
# kernel trace + stats of the TRAIN STEP ONLY (bench.py --no-breakdown --no-eval --cpu-steps 0: no roofline replays, no event-timed
# extra steps, no evaluation leg -> the per-step table is a true breakdown), then three PMC passes (SQ, FETCH_SIZE, WRITE_SIZE) as
# MI355X_MICROARCH.md prescribes, then the summaries: <tag>_kernel_stats_<cfg>.{csv,txt}, <tag>_pmc_<cfg>.txt,
# <tag>_step_sequence_<cfg>.txt, <tag>_stepbytes_<cfg>.json under gpurun_out/<tag>_<cfg>/summary/ (copy them into profiles/).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=ml1m
for a in "$@"; do if [ "$prev" = "--config" ]; then cfg=$a; fi; prev=$a; done
out=$root/gpurun_out/${tag}_${cfg}   # one directory per (tag, configuration): the summary scripts take the only trace they find
rm -rf $out
mkdir -p $out/summary
common="--no-breakdown --no-eval --cpu-steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 20 --warmup 5 $common "$@" > $out/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $root/bench.py --steps 4 --warmup 2 $common "$@" > $out/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 4 --warmup 2 $common "$@" > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 4 --warmup 2 $common "$@" > $out/pmc_write.log 2>&1 || exit 1
cd $root
python3 - > $out/summary/${tag}_build_${cfg}.json <<PY
import hashlib, json, os, subprocess
lib = os.environ.get("B4R_LIB_PATH") or "bert4rec_amd/libb4r_hip.so"
print(json.dumps({"lib": lib, "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]}))
PY
cp $out/trace/*/*_kernel_stats.csv $out/summary/${tag}_kernel_stats_${cfg}.csv
python3 tools/stats.py $out/trace 25 40 > $out/summary/${tag}_kernel_stats_${cfg}.txt
python3 tools/seq.py $out/trace > $out/summary/${tag}_step_sequence_${cfg}.txt
python3 tools/pmc.py gpurun_out/${tag}_${cfg} 40 > $out/summary/${tag}_pmc_${cfg}.txt
python3 tools/stepbytes.py $out $cfg $tag > $out/summary/${tag}_stepbytes_${cfg}.json
python3 $root/bench.py --steps 200 --warmup 30 "$@" > $out/summary/${tag}_bench_${cfg}.json 2> $out/bench.err
cat $out/summary/${tag}_kernel_stats_${cfg}.txt $out/summary/${tag}_pmc_${cfg}.txt $out/summary/${tag}_stepbytes_${cfg}.json
