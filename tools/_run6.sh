cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/s6_gputests_full.log
tail -4 gpurun_out/s6_gputests_full.log
NELLIE_SWITCH_INTERVAL=0.0001 python tools/bench_stream_lanes.py 64 1 2 2>/dev/null | tail -1 > gpurun_out/s6_lanes.txt; cat gpurun_out/s6_lanes.txt
python tools/bench_stream_lanes.py 64 1 2 2>/dev/null | tail -1 >> gpurun_out/s6_lanes.txt; tail -1 gpurun_out/s6_lanes.txt
