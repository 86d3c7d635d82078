#!/bin/bash
# A/B of library variants (tools/build_variant.sh NAME "-D...") through the bench line, twice, alternating: tools/ab_bench_variants.sh [NAME ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in ${@:-base}; do
  NELLIE_HIP_LIB=$R/nellie_amd/variants/libnellie_hip_$v.so python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-io 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['groups'].items()})"
done; done
