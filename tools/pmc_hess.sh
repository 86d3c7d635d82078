#!/bin/bash
# SQ counters of the Hessian walk (several --pmc passes, counters only): tools/pmc_hess.sh [Z Y X] -> gpurun_out/r3/pmc_hess.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/${RND:-r6}; OUT=$R/gpurun_out/${RND:-r6}/pmc_sq_${TAG:-x}.txt; : > $OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  rm -rf /tmp/pmch_$i && rocprofv3 --pmc $C --output-format csv -d /tmp/pmch_$i -- python $R/tools/prof_filter.py ${1:-512} ${2:-1024} ${3:-1024} 1 > /tmp/pmch_$i.log 2>&1
  F=$(find /tmp/pmch_$i -name '*counter_collection.csv' | head -1)
  python - "$F" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); rows = []
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:44]
    if 'hessian' not in k and 'vesselness_queue' not in k and 'gauss_zyx' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k, {c: "%.4g" % (v / n[(k, c)]) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
PY
  tail -1 /tmp/pmch_$i.log | cut -c1-160 >> $OUT
done
cat $OUT
