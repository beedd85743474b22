set -o pipefail
cd $GRAFT_REPO_ROOT
for v in abprof ab4; do
B4R_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libb4r_$v.so timeout -k 10 300 python tools/bench_attn_block.py 2>&1 | grep -E "attention block|sweep|end"
done
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
for i in 1 2 3; do python bench.py --no-eval --no-breakdown --steps 200 --warmup 30 2>&1 | tail -1 | cut -c1-110; done
