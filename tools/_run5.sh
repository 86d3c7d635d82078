cd $GRAFT_REPO_ROOT
for cfg in "NELLIE_MK_PEAK4=0" "A=1"; do echo "[$cfg]" >> gpurun_out/s5_markers_ab.txt; env $cfg python tools/bench_markers.py 2>/dev/null | tail -1 >> gpurun_out/s5_markers_ab.txt; done
cat gpurun_out/s5_markers_ab.txt
NCCL_DEBUG=INFO NCCL_DEBUG_FILE=$GRAFT_REPO_ROOT/gpurun_out/s5_nccl_%p.log python -m pytest tests -m gpu -q -rf 2>&1 | tail -80 > gpurun_out/s5_gputests_full.log
tail -6 gpurun_out/s5_gputests_full.log
ls -la gpurun_out/s5_nccl_* | tail -3
