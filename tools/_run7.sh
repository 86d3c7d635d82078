cd $GRAFT_REPO_ROOT
python tools/fuzz_parity.py 240 31 gpurun_out/s7_fuzz_parity_small_seed31.txt > /dev/null 2>&1; tail -1 gpurun_out/s7_fuzz_parity_small_seed31.txt | cut -c1-400
python tools/fuzz_parity.py 200 32 gpurun_out/s7_fuzz_parity_big_seed32.txt big > /dev/null 2>&1; tail -1 gpurun_out/s7_fuzz_parity_big_seed32.txt | cut -c1-400
python tools/fuzz_stages.py 200 33 gpurun_out/s7_fuzz_stages_seed33.txt > /dev/null 2>&1; tail -1 gpurun_out/s7_fuzz_stages_seed33.txt | cut -c1-400
python tools/fuzz_slabs.py 150 34 gpurun_out/s7_fuzz_slabs_seed34.txt > /dev/null 2>&1; tail -1 gpurun_out/s7_fuzz_slabs_seed34.txt | cut -c1-400
python tools/fuzz_api.py 150 35 gpurun_out/s7_fuzz_api_seed35.txt > /dev/null 2>&1; tail -1 gpurun_out/s7_fuzz_api_seed35.txt | cut -c1-400
