#!/bin/bash
# per-kernel time of one Filter+Label pass at a given frame size: tools/kstats.sh Z Y X [reps]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${4:-4}
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/tools/prof_filter.py ${1:-128} ${2:-512} ${3:-512} $N > /tmp/ks.log 2>&1
F=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
python - "$F" "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per pass %.3f, launches per pass %.0f" % (tot / n / 1e6, sum(int(r["Calls"]) for r in rows) / n))
for r in rows[:34]:
    print("%-64s calls %6.1f  ms %7.3f  avg us %8.1f" % (r["Name"][:64], int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6, float(r["AverageNs"]) / 1e3))
PY
