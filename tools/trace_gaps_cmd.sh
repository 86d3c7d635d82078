#!/bin/bash
# GPU idle time of the LAST repetition of any driver script (kernel trace): tools/trace_gaps_cmd.sh OUT.txt REPS script args...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; REPS=$2; shift 2
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python "$@" > /tmp/kt.log 2>&1
F=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$F" "$REPS" <<'PY' > $R/gpurun_out/$OUT
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
reps = int(sys.argv[2])
rows = rows[len(rows) - len(rows) // reps:]           # the last repetition, by kernel count
span = (max(e for _, e, _ in rows) - rows[0][0]) / 1e6
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
prev_name = rows[0][2]
t_origin = rows[0][0]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e6, (cur_e - t_origin) / 1e6, prev_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_name = n
busy += cur_e - cur_s
print(f"span {span:.3f} ms  busy {busy/1e6:.3f} ms  idle {span - busy/1e6:.3f} ms  kernels {len(rows)}")
for g in sorted(gaps, reverse=True)[:40]:
    print(f"  gap {g[0]:.3f} ms at {g[1]:7.3f} after {g[2]} before {g[3]}")
PY
cat $R/gpurun_out/$OUT
