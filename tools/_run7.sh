set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 bash tools/prof.sh r02_e > gpurun_out/prof_r02_e.log 2>&1 || echo "ml1m prof failed"
timeout -k 10 300 bash tools/prof.sh r02_e --config steam > gpurun_out/prof_r02_e_steam.log 2>&1 || echo "steam prof failed"
timeout -k 10 400 bash tools/prof.sh r02_e --config ml20m_4l > gpurun_out/prof_r02_e_ml20m.log 2>&1 || echo "ml20m prof failed"
python bench.py --no-eval --cpu-steps 0 --steps 200 --warmup 30 --ragged > gpurun_out/r02_e_bench_ml1m_ragged.json 2>/dev/null
python bench.py --no-eval --cpu-steps 0 --steps 200 --warmup 30 --ragged --bucketed > gpurun_out/r02_e_bench_ml1m_ragged_bucketed.json 2>/dev/null
tail -c 400 gpurun_out/r02_e_bench_ml1m_ragged_bucketed.json
ls gpurun_out/r02_e_ml1m/summary gpurun_out/r02_e_steam/summary gpurun_out/r02_e_ml20m_4l/summary | head -30
