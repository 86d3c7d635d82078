#!/bin/bash
# per-kernel time of any driver script: tools/kstats_cmd.sh REPS script args...   (REPS = passes the script makes, for the per-pass figures)
cd /tmp && export TMPDIR=/tmp
N=$1; shift
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python "$@" > /tmp/ks.log 2>&1
F=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
python - "$F" "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per pass %.3f, launches per pass %.0f" % (tot / n / 1e6, sum(int(r["Calls"]) for r in rows) / n))
for r in rows[:60]:
    print("%-64s calls %6.1f  ms %7.3f  avg us %8.1f" % (r["Name"][:64], int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6, float(r["AverageNs"]) / 1e3))
PY
