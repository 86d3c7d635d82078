#!/bin/bash
# A/B of environment switches through the bench line, twice, alternating: tools/ab_env.sh "A=1" "A=2 B=3" ...  ("-" = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for cfg in "$@"; do
  e=(); [ "$cfg" != "-" ] && e=($cfg)
  env "${e[@]}" python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-io 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['groups'].items()})"
done; done
