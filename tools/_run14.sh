cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_hip_full_size.py -m gpu -q -x -k "held_back or chain or 1024 or full or crops" 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
for rep in 1 2 3; do
for cfg in "NELLIE_RESOLVE_DEFER=0" "A=1" "NELLIE_RESOLVE_DEFER=4"; do
  env $cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-io 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['groups'].items()})" >> gpurun_out/s14_ab_resolve_defer.txt
done; done
cat gpurun_out/s14_ab_resolve_defer.txt
NELLIE_RESOLVE_DEFER=5 python tools/fuzz_parity.py 120 53 gpurun_out/s14_fuzz_parity_defer5_seed53.txt > /dev/null 2>&1; tail -1 gpurun_out/s14_fuzz_parity_defer5_seed53.txt | cut -c1-300
