#!/bin/bash
# per-launch kernel durations of one Filter+Label pass (rocprofv3 kernel trace) -> gpurun_out/launches.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/tools/prof_filter.py ${1:-1024} ${2:-1024} ${3:-1024} 2 > /tmp/kt.log 2>&1
F=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY' > $R/gpurun_out/launches.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
half = len(rows)//2
for r in rows[half:]:
    d = (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    if d > 0.03:
        print(f"{d:8.3f} ms  {r['Kernel_Name'][:110]}  vgpr={r.get('VGPR_Count','?')} lds={r.get('LDS_Block_Size','?')} wg={r.get('Workgroup_Size','?')}")
PY
tail -3 /tmp/kt.log
