cd $GRAFT_REPO_ROOT
for r in 4 3 5; do for v in 0 1; do echo "== LDS64=$v R=$r" >> gpurun_out/s8_ubench_lds64.txt; tools/ubench/gzyx_fused_lds$v 1024 1024 1024 $r 2>&1 | grep -i "LIBRARY\|two kernels\|equal\|differ" >> gpurun_out/s8_ubench_lds64.txt; done; done
cat gpurun_out/s8_ubench_lds64.txt
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "gaussian or fused_cascade or 2d or golden" 2>&1 | tail -3
for rep in 1 2; do
for lib in "" "nellie_amd/variants/libnellie_hip_lds32.so"; do
  NELLIE_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-io 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$lib]', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['groups'].items()})" >> gpurun_out/s8_ab_lds64.txt
done; done
cat gpurun_out/s8_ab_lds64.txt
python tools/bench_2d.py 2>/dev/null | tail -1 > gpurun_out/s8_bench_2d.txt; cat gpurun_out/s8_bench_2d.txt
bash tools/kstats_cmd.sh 54 $GRAFT_REPO_ROOT/tools/bench_2d.py > gpurun_out/s8_kstats_2d.txt 2>&1; head -30 gpurun_out/s8_kstats_2d.txt
python tools/fuzz_stages.py 230 33 gpurun_out/s8_fuzz_stages_seed33.txt > /dev/null 2>&1; tail -1 gpurun_out/s8_fuzz_stages_seed33.txt | cut -c1-400; grep -c "both raise" gpurun_out/s8_fuzz_stages_seed33.txt
