for g in 0 640 768 896 960 1280 2048; do
  echo "== grid_s $g"
  MMM_EXP_GRID_S=$g python3 tools/shard_run.py 5 1 40 2>&1 | grep "ms per step" | cut -c1-260
done
for g in 0 768 782 896 1563; do
  echo "== cfg4 grid_s $g"
  MMM_EXP_GRID_S=$g python3 tools/shard_run.py 4 1 40 2>&1 | grep "ms per step" | cut -c1-260
done
