cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_sharded.py -m gpu -x -q -k "test_rccl_communicator_world1" 2>&1 | tail -60 > gpurun_out/s2_rccl_fail.log
python -m pytest tests/test_hip_parity.py tests/test_hip_full_size.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/s2_parity_e28.log
for rep in 1 2; do
for lib in "" "nellie_amd/variants/libnellie_hip_e32.so"; do
  NELLIE_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-io 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$lib]', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['groups'].items()})" >> gpurun_out/s2_ab_e28.txt
done; done
bash tools/kstats.sh 128 512 512 8 > gpurun_out/s2_kstats_c5.txt 2>&1
cat gpurun_out/s2_ab_e28.txt; tail -3 gpurun_out/s2_parity_e28.log; tail -5 gpurun_out/s2_rccl_fail.log
