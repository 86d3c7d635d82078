cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "chain or golden" 2>&1 | tail -4 > gpurun_out/s4_chain_tests.log
tail -2 gpurun_out/s4_chain_tests.log
for rep in 1 2; do
for cfg in "NELLIE_CHAIN_AHEAD=0" "A=1"; do
  env $cfg python bench.py --steps 20 --warmup 3 --shape 128 512 512 --no-cpu-baseline --no-io 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg] frame 128x512x512', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['groups'].items()})" >> gpurun_out/s4_ab_c5_ahead.txt
  env $cfg python tools/bench_stream.py 32 128 512 512 2>/dev/null | tail -1 >> gpurun_out/s4_ab_c5_ahead.txt
done; done
cat gpurun_out/s4_ab_c5_ahead.txt
for cfg in "NELLIE_CHAIN_AHEAD=0" "A=1"; do env $cfg bash tools/trace_gaps.sh 128 512 512 > /dev/null 2>&1; cp gpurun_out/gaps.txt gpurun_out/s4_gaps_c5_$cfg.txt; head -3 gpurun_out/gaps.txt; done
python tools/bench_2d.py 2>/dev/null | tail -3 > gpurun_out/s4_bench_2d.txt; cat gpurun_out/s4_bench_2d.txt
NCCL_DEBUG=INFO NCCL_DEBUG_FILE=$GRAFT_REPO_ROOT/gpurun_out/s4_nccl_%p.log python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/s4_gputests_full.log
tail -3 gpurun_out/s4_gputests_full.log
tail -30 gpurun_out/s4_nccl_*.log | grep -i -B2 -A2 "warn\|error\|fail" | head -60
rocm-smi --showmeminfo vram 2>/dev/null | head -8
