cd $GRAFT_REPO_ROOT
NELLIE_RESOLVE_DEFER=2 python tools/fuzz_parity.py 150 51 gpurun_out/s13_fuzz_parity_defer2_seed51.txt > /dev/null 2>&1; tail -1 gpurun_out/s13_fuzz_parity_defer2_seed51.txt | cut -c1-300
NELLIE_RESOLVE_DEFER=2 NELLIE_CHAIN_AHEAD=0 python tools/fuzz_parity.py 120 52 gpurun_out/s13_fuzz_parity_big_defer2_seed52.txt big > /dev/null 2>&1; tail -1 gpurun_out/s13_fuzz_parity_big_defer2_seed52.txt | cut -c1-300
PFX=r05 bash tools/make_profiles.sh > gpurun_out/s13_make_profiles.log 2>&1; tail -c 300 gpurun_out/r05_bench_n1_1024cube.json
python tools/bench_markers.py 2>/dev/null | tail -1 > gpurun_out/s13_markers.json; cat gpurun_out/s13_markers.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > gpurun_out/s13_gputests_full.log; cat gpurun_out/s13_gputests_full.log
