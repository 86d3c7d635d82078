#!/bin/bash
# Produces the files committed under profiles/ (run on the GPU box through gpurun; results land in gpurun_out/):
#   ${PFX}_bench_n1_1024cube.json       the default bench line
#   ${PFX}_kernel_stats_1024cube.csv    rocprofv3 --kernel-trace --stats summary of the same bench command
#   ${PFX}_pmc_hbm_bytes_1024cube.json  HBM bytes per kernel launch: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
#                                    (KB units; FETCH_SIZE is doubled on gfx950, see MI355X_MICROARCH.md)
#   ${PFX}_zslab_world1_128x2048x2048.json  one rank's share of the 8-GPU Z-slab run of BASELINE config 4 (the bench's Z-slab child at
#                                    world 1 over a real one-rank RCCL communicator): what a rank costs before any neighbour exists
R=${GRAFT_REPO_ROOT:-/root/repo}
PFX=${PFX:-r05}
cd /tmp && export TMPDIR=/tmp
if [ "$1" != "nobench" ]; then python $R/bench.py --steps 10 --warmup 2 > /tmp/bench.out 2>/tmp/bench.err; tail -1 /tmp/bench.out > $R/gpurun_out/${PFX}_bench_n1_1024cube.json; fi
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-io > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${PFX}_kernel_stats_1024cube.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C && rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python $R/tools/prof_filter.py 1024 1024 1024 1 > /tmp/pmc_$C.log 2>&1
done
python - $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) > $R/gpurun_out/${PFX}_pmc_hbm_bytes_1024cube.json <<'PY'
import csv, sys, json, collections
N = 1024 ** 3
def load(path):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k] += float(r['Counter_Value']); n[k] += 1
    return acc, n
f, nf = load(sys.argv[1]); w, nw = load(sys.argv[2])
out = []
for k in sorted(f, key=lambda k: -f[k]):
    L = nf[k]
    fk, wk = f[k] / L, w.get(k, 0.0) / max(1, nw.get(k, 1))
    out.append({"kernel": k, "launches": L, "fetch_size_kb_per_launch": fk, "write_size_kb_per_launch": wk,
                "fetch_bytes_per_voxel_raw": fk * 1024 / N, "fetch_bytes_per_voxel_corrected_x2": 2 * fk * 1024 / N,
                "write_bytes_per_voxel": wk * 1024 / N})
json.dump(out, sys.stdout, indent=1)
PY
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python $R/bench.py --zslab-child --gpus 1 --steps 10 --warmup 2 2>/tmp/zs.err | tail -1 > $R/gpurun_out/${PFX}_zslab_world1_128x2048x2048.json
tail -2 /tmp/ks.log; tail -c 600 $R/gpurun_out/${PFX}_bench_n1_1024cube.json
# r03 additions: the SAME 8-slab volume (BASELINE config 4) as 8 slab contexts on this one GPU over the loopback transport,
# kernel stats of the slab step at world 1, GPU idle time of a config-5 frame with and without the device-resident chain
cd $R
python bench.py --zslab-on-one-gpu 8 --steps 2 --warmup 1 2>/tmp/z1.err | tail -1 > gpurun_out/${PFX}_c4_as_8_slabs_on_one_gpu.json
cd /tmp
rm -rf /tmp/ksz && OMP_NUM_THREADS=1 NELLIE_BENCH_CLEAN_EXIT=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksz -- python $R/bench.py --zslab-child --gpus 1 --steps 4 --warmup 2 > /tmp/ksz.log 2>&1
python - $(find /tmp/ksz -name '*kernel_stats.csv' | head -1) > $R/gpurun_out/${PFX}_kernel_stats_zslab_world1_128x2048x2048.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print(f"all kernels of the run (2 equality + 2 warm-up + 4 timed + 2 serial-group steps of one 128x2048x2048 slab): {tot/1e6:.1f} ms, {calls} launches")
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:45]:
    print(f"{r['Name'][:64]:64s} calls {int(r['Calls']):6d}  ms {float(r['TotalDurationNs'])/1e6:9.3f}  avg us {float(r['AverageNs'])/1e3:9.1f}")
PY
cd $R
tools/trace_gaps.sh 128 512 512 > /dev/null 2>&1; cp gpurun_out/gaps.txt gpurun_out/${PFX}_gaps_c5_frame_chain.txt
NELLIE_DEVICE_CHAIN=0 tools/trace_gaps.sh 128 512 512 > /dev/null 2>&1; cp gpurun_out/gaps.txt gpurun_out/${PFX}_gaps_c5_frame_sync.txt
tools/trace_gaps.sh 1024 1024 1024 > /dev/null 2>&1; cp gpurun_out/gaps.txt gpurun_out/${PFX}_gaps_1024cube_chain.txt
