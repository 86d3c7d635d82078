#!/bin/bash
# duration of every launch of the walk / resolve kernels of ONE Filter+Label pass, in launch order: tools/walk_launches.sh Z Y X
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/tools/prof_filter.py $1 $2 $3 2 > /tmp/kt.log 2>&1
F=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY'
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))))
for pat in ("hessian_", "vesselness_queue"):
    d = [round((e - s) / 1e3, 1) for s, e, n in rows if pat in n]
    print(pat, "us per launch (two passes):", d)
PY
