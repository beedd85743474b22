set -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-eval --no-breakdown --steps 200 --warmup 30 2>&1 | tail -1 | cut -c1-110; done
python bench.py --no-eval --no-breakdown --steps 200 --warmup 30 --graph 2>&1 | tail -1 | cut -c1-110
export B4R_SIDE_STREAM=4
for i in 1 2; do python bench.py --no-eval --no-breakdown --steps 200 --warmup 30 2>&1 | tail -1 | cut -c1-110; done
for i in 1 2; do python bench.py --no-eval --no-breakdown --steps 200 --warmup 30 --graph 2>&1 | tail -1 | cut -c1-110; done
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
