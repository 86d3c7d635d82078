cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "pack or stream or c5 or run_on_disk or ome_tiff" 2>&1 | tail -4
for rep in 1 2 3; do
for cfg in "NELLIE_STREAM_PACK_WITH_LABEL=0" "A=1"; do
  echo "[$cfg]" >> gpurun_out/s11_ab_pack_with_label.txt
  env $cfg python tools/bench_stream.py 64 128 512 512 2>/dev/null | tail -1 >> gpurun_out/s11_ab_pack_with_label.txt
done; done
cat gpurun_out/s11_ab_pack_with_label.txt
