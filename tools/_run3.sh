cd $GRAFT_REPO_ROOT
NELLIE_HV_NP=2 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/s3_parity_np2.log
tail -3 gpurun_out/s3_parity_np2.log
for rep in 1 2; do
for cfg in "A=1" "NELLIE_HV_NP=2"; do
  env $cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-io 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['groups'].items()})" >> gpurun_out/s3_ab_np2.txt
done; done
cat gpurun_out/s3_ab_np2.txt
python -m pytest tests -m gpu -q -rf 2>&1 | tail -120 > gpurun_out/s3_gputests_full.log
tail -5 gpurun_out/s3_gputests_full.log
