set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_gpu_model.py tests/test_gpu_distributed.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
for i in 1 2 3; do python bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 200 --warmup 30 2>&1 | tail -1 | cut -c1-120; done
python bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 200 --warmup 30 --force-dist 2>&1 | tail -1 | cut -c1-120
python bench.py --no-eval --no-breakdown --cpu-steps 0 --steps 200 --warmup 30 --force-dist --graph 2>&1 | tail -1 | cut -c1-120
