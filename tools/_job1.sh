set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -k "fp16 or f32tol" -x -q > $O/pytest_f16.log 2>&1 || { tail -30 $O/pytest_f16.log; exit 1; }
tail -2 $O/pytest_f16.log
V=$PWD/yolo-fpga-accelerator_amd/build/lib_c0noxcd.so
for r in 1 2; do
  echo "== default round $r"; python3 tools/f16_layers.py 128 10 2>/dev/null > $O/layers_def_$r.txt; grep -E "^L 0 |^L13|^L15|^L19|^L21|^sum" $O/layers_def_$r.txt
  echo "== c0 launch order (no XCD remap) round $r"; YOLO2_HIP_LIB=$V python3 tools/f16_layers.py 128 10 2>/dev/null > $O/layers_c0noxcd_$r.txt; grep -E "^L 0 |^sum" $O/layers_c0noxcd_$r.txt
  echo "== ring_sq round $r"; YOLO2_F16_RING_SQ=1 python3 tools/f16_layers.py 128 10 2>/dev/null > $O/layers_ringsq_$r.txt; grep -E "^L13|^L15|^L19|^L21|^sum" $O/layers_ringsq_$r.txt
done
timeout -k 10 400 bash tools/f16_traffic.sh r4b > $O/traffic.log 2>&1; head -8 gpurun_out/f16_traffic_r4b_summary.txt
