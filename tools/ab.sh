#!/bin/bash
# A/B: bench each variants/*.so in turn (experiment helper)
for v in variants/*.so; do
  cp $v nellie_amd/libnellie_hip.so
  echo "== $v"
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"groups_ms_per_step\"])"
done
