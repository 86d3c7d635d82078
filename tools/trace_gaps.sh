#!/bin/bash
# GPU idle time inside one Filter+Label pass (kernel trace): span, union of busy intervals, the largest gaps
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/tools/prof_filter.py ${1:-1024} ${2:-1024} ${3:-1024} 2 > /tmp/kt.log 2>&1
F=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY' > $R/gpurun_out/gaps.txt
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# second repetition = after the last convert/first gauss of rep 2: take the second half by count
half = len(rows) // 2
rows = rows[half:]
span = (max(e for _, e, _ in rows) - rows[0][0]) / 1e6
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
prev_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e6, prev_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_name = n
busy += cur_e - cur_s
print(f"span {span:.3f} ms  busy {busy/1e6:.3f} ms  idle {span - busy/1e6:.3f} ms  kernels {len(rows)}")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  gap {g[0]:.3f} ms after {g[1]} before {g[2]}")
PY
cat $R/gpurun_out/gaps.txt
