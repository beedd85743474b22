set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
for i in 1 2 3; do python bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 200 --warmup 30 2>&1 | tail -1 | cut -c1-120; done
B4R_EMB_FUSED=0 python bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 200 --warmup 30 2>&1 | tail -1 | cut -c1-120
(cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/trv_ && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trv_ -- python3 $GRAFT_REPO_ROOT/bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/trv.log 2>&1)
python tools/seq.py gpurun_out/trv_ | grep -E "zero2|sum|attn_block_fwd|ln_fwd"
