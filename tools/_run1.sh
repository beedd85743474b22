# scratch driver for one GPU session: the parity suites, then bench + one-step kernel sequence per library variant ("" = in-tree build)
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_blocks.py -x -q -m gpu > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export B4R_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libb4r_$v.so; fi
  for i in 1 2; do python bench.py --no-eval --no-breakdown --steps 200 --warmup 30 2>&1 | tail -1 | cut -c1-110; done
  (cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/trv_$v && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trv_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/trv.log 2>&1)
  python tools/seq.py gpurun_out/trv_$v | grep -E "${GREP:-slab_reduce|^sum}"
done
