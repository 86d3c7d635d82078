#!/bin/bash
# Timeline of L streaming lanes (kernel + memory-copy trace of tools/diag_stream_lanes.py's configurations): tools/trace_lanes.sh OUT.txt [T]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; T=${2:-16}
for cfg in "1 0 f32" "2 0 f32" "3 0 f32"; do
rm -rf /tmp/kt && NELLIE_DIAG_ONLY="$cfg" rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt -- python $R/tools/diag_stream_lanes.py $T > /tmp/kt.log 2>&1
tail -2 /tmp/kt.log
K=$(find /tmp/kt -name '*kernel_trace.csv' | head -1); M=$(find /tmp/kt -name '*memory_copy_trace.csv' | head -1)
python - "$K" "$M" <<'PY'
import csv, sys, collections
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:48], r.get('Queue_Id', '?')) for r in csv.DictReader(open(sys.argv[1]))]
ms = list(csv.DictReader(open(sys.argv[2])))
ks.sort()
half = ks[len(ks) // 2:]                     # the timed pass
t0, t1 = half[0][0], max(e for _, e, _, _ in half)
span = (t1 - t0) / 1e6
# union of kernel time, and time with >= 2 kernels running
ev = []
for s, e, _, _ in half: ev += [(s, 1), (e, -1)]
ev.sort()
busy = over = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
print(f"timed pass: span {span:.2f} ms, some kernel running {busy/1e6:.2f} ms, two or more {over/1e6:.2f} ms, kernels {len(half)}, queues {sorted(set(q for *_, q in half))}")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in half: by[n][0] += 1; by[n][1] += e - s
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {n:48s} calls {c:5d}  total ms {d/1e6:8.3f}  avg us {d/c/1e3:8.1f}")
print("memory copies in the timed pass:")
cols = ms[0].keys() if ms else []
agg = collections.defaultdict(lambda: [0, 0, 0])
for r in ms:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t0 or s > t1: continue
    key = (r.get('Direction', '?'), 'big' if (e - s) > 200000 else 'small')
    agg[key][0] += 1; agg[key][1] += e - s
for k, (c, d, _) in sorted(agg.items()):
    print(f"  {k}: {c} copies, total {d/1e6:.2f} ms, avg {d/c/1e3:.1f} us")
big = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in ms if t0 <= int(r['Start_Timestamp']) <= t1 and r.get('Direction', '').endswith('HOST_TO_DEVICE') and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 200000)
if len(big) > 1:
    gaps = [(b[0] - a[1]) / 1e6 for a, b in zip(big, big[1:])]
    print("  uploads: durations ms", [round((e - s_) / 1e6, 2) for s_, e in big])
    print("  uploads: gap to the next one ms", [round(g, 2) for g in gaps])
PY
done > $R/gpurun_out/$OUT 2>&1
cat $R/gpurun_out/$OUT
