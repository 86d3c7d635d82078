#!/bin/bash
# Every GPU operation (kernels AND the runtime's blit copies / fills) of the LAST step of a one-rank Z-slab run, in start order:
#   tools/trace_slab_timeline.sh OUT.txt [Z Y X] [reps]      (env is passed through, e.g. NELLIE_DEVICE_CHAIN_SLABS=1)
# Answers "what are the ~160 __amd_rocclr_copyBuffer of a slab step": each one is listed with its duration, its stream (queue)
# and the kernels around it.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; Z=${2:-128}; Y=${3:-2048}; X=${4:-2048}; REPS=${5:-3}
rm -rf /tmp/kt && NELLIE_PROF_CALLS=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/tools/prof_slab.py $Z $Y $X $REPS > /tmp/kt.log 2>&1
tail -3 /tmp/kt.log
F=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$F" "$REPS" <<'PY' > $R/gpurun_out/$OUT
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:56], r.get('Queue_Id', '?'),
                 int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)))
rows.sort()
reps = int(sys.argv[2])
# the last step starts at the last launch of the first Gaussian kernel group: find the last 'gauss_march' whose predecessor is not a gauss
starts = [i for i, r in enumerate(rows) if r[2].startswith('void gauss_march') or r[2].startswith('gauss_march')]
# 5 Z passes per step -> the 5th from the end begins the last step
first = starts[-5] if len(starts) >= 5 else 0
# walk back over the step's prologue (fills / small kernels issued by filter_begin before the first Gaussian)
while first > 0 and rows[first][0] - rows[first - 1][1] < 200_000 and not rows[first - 1][2].startswith('rl_paint'):
    first -= 1
step = rows[first:]
t0 = step[0][0]
span = (max(e for _, e, *_ in step) - t0) / 1e6
by = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q, g in step:
    by[n][0] += 1; by[n][1] += (e - s) / 1e6
print(f"last step: {len(step)} operations, span {span:.3f} ms, sum of durations {sum(v[1] for v in by.values()):.3f} ms")
for n, (c, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:56s} x{c:4d}  {ms:8.3f} ms  avg {ms / c * 1e3:8.1f} us")
print("\ntimeline (start ms, dur us, queue, grid, name):")
prev_e = t0
for s, e, n, q, g in step:
    gap = (s - prev_e) / 1e3
    print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:9.1f} q{q:>3s} g{g:>10d} {n}" + (f"   [idle {gap:.0f} us before]" if gap > 20 else ""))
    prev_e = max(prev_e, e)
PY
head -60 $R/gpurun_out/$OUT
