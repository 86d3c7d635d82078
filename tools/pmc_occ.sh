#!/bin/bash
# average occupancy / busy counters per kernel: tools/pmc_occ.sh [Z Y X]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc_occ && NELLIE_RESOLVE_SERIAL=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_occ -- python $R/tools/prof_filter.py ${1:-1024} ${2:-1024} ${3:-1024} 1 > /tmp/pmc_occ.log 2>&1
F=$(find /tmp/pmc_occ -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:48]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:12]:
    gui = v.get('GRBM_GUI_ACTIVE', 0)
    print("%-48s waves %.3g  wave_cycles %.3g  busy %.3g  gui_active %.3g  avg waves resident ~ %.1f" % (k, v.get('SQ_WAVES', 0), v.get('SQ_WAVE_CYCLES', 0), v.get('SQ_BUSY_CYCLES', 0), gui, v.get('SQ_WAVE_CYCLES', 0) / gui if gui else 0))
PY
tail -2 /tmp/pmc_occ.log | cut -c1-200
