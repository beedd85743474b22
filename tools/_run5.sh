set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
for i in 1 2; do python bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 200 --warmup 30 --ragged 2>&1 | tail -1 | cut -c1-200; done
for i in 1 2; do python bench.py --no-eval --cpu-steps 0 --steps 200 --warmup 30 --ragged --bucketed 2>&1 | tail -1 | cut -c1-260; done
python bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 200 --warmup 30 2>&1 | tail -1 | cut -c1-200
