#!/bin/bash
# A/B of the walk inside the bench step: tools/ab_walk.sh "ENV=.. ENV=.." ...   (one bench run per configuration, same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  e=(); [ "$cfg" != "-" ] && e=($cfg)
  env "${e[@]}" python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-io --no-pmc ${SHAPE:+--shape $SHAPE} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['roofline']['groups']; print('[$cfg]', d['ms_per_step'], 'walk', g['vesselness']['ms_avg'], 'resolve', g['vesselness_resolve']['ms_avg'], {k:v['ms_per_step'] for k,v in g.items()})"
done
