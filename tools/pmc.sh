#!/bin/bash
# rocprofv3 PMC pass (counters only, no tracing): tools/pmc.sh "<COUNTERS>" <tag> [nz ny nx] -> gpurun_out/pmc_<tag>.txt
# (per-kernel sums of each counter over one Filter+Label pass)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
C="$1"; T="$2"; shift 2
rm -rf /tmp/pmc_$T && rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$T -- python $R/tools/prof_filter.py ${1:-512} ${2:-1024} ${3:-1024} 1 > /tmp/pmc_$T.log 2>&1
F=$(find /tmp/pmc_$T -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY' > $R/gpurun_out/pmc_$T.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    n[(k, r['Counter_Name'])] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print(k, {c: (round(v), n[(k, c)]) for c, v in acc[k].items()})
PY
tail -2 /tmp/pmc_$T.log
