R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "-" "NELLIE_GAUSS_FUSED=0" "NELLIE_ZYX_MIN_WGS=256" "NELLIE_ZYX_MIN_WGS=1024" "NELLIE_ZYX_MIN_WGS=2048"; do
  e=(); [ "$cfg" != "-" ] && e=($cfg)
  env "${e[@]}" python $R/bench.py --steps 10 --warmup 2 --shape 128 512 512 --no-cpu-baseline --no-io 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['groups'].items()})"
done
