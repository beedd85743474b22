set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py --steps 200 --warmup 30 > gpurun_out/final/r02_e_bench_ml1m.json 2> gpurun_out/final/bench_ml1m.err
python bench.py --steps 200 --warmup 30 --config steam > gpurun_out/final/r02_e_bench_steam.json 2> gpurun_out/final/bench_steam.err
python bench.py --steps 50 --warmup 10 --config ml20m_4l > gpurun_out/final/r02_e_bench_ml20m_4l.json 2> gpurun_out/final/bench_ml20m.err
tail -c 200 gpurun_out/final/r02_e_bench_ml20m_4l.json
