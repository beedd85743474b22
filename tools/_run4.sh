set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py --steps 200 --warmup 30 > gpurun_out/final/r02_c_bench_ml1m.json 2> gpurun_out/final/bench_ml1m.err && tail -c 300 gpurun_out/final/r02_c_bench_ml1m.json
python bench.py --steps 200 --warmup 30 --config steam > gpurun_out/final/r02_c_bench_steam.json 2> gpurun_out/final/bench_steam.err
python bench.py --steps 50 --warmup 10 --config ml20m_4l > gpurun_out/final/r02_c_bench_ml20m_4l.json 2> gpurun_out/final/bench_ml20m.err
python bench.py --steps 200 --warmup 30 --ragged --no-eval --cpu-steps 0 > gpurun_out/final/r02_c_bench_ml1m_ragged.json 2> gpurun_out/final/bench_ragged.err
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1; tail -3 gpurun_out/final/gpu_tests.log
for e in bert4rec_steam_example bert4rec_ml_20m_example bert4rec_evaluation_example; do timeout -k 10 300 python examples/$e.py > gpurun_out/final/$e.log 2>&1; echo "$e rc=$?"; tail -2 gpurun_out/final/$e.log; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
