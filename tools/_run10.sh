cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/s10_gputests_full.log; tail -2 gpurun_out/s10_gputests_full.log
python tools/bench_2d.py 2>/dev/null | tail -1 > gpurun_out/s10_bench_2d.txt; cat gpurun_out/s10_bench_2d.txt
NELLIE_GAUSS_FUSED=1 NELLIE_CHAIN_AHEAD=1 python tools/fuzz_parity.py 200 41 gpurun_out/s10_fuzz_parity_fused_ahead_seed41.txt > /dev/null 2>&1; tail -1 gpurun_out/s10_fuzz_parity_fused_ahead_seed41.txt | cut -c1-400
python tools/fuzz_files.py 150 42 gpurun_out/s10_fuzz_files_seed42.txt > /dev/null 2>&1; tail -1 gpurun_out/s10_fuzz_files_seed42.txt | cut -c1-400
python tools/fuzz_stages.py 120 43 gpurun_out/s10_fuzz_stages_seed43.txt > /dev/null 2>&1; tail -1 gpurun_out/s10_fuzz_stages_seed43.txt | cut -c1-400
