mkdir -p gpurun_out/r3k
for R in 0 64 256; do
  for mode in 1 4 0; do
    MMM_CTM_CPL=$mode python tools/bench_ctm.py --config 3 --restarts $R --steps 30 2>/dev/null | tail -1 | sed "s/^/R=$R cpl=$mode /"
  done
done
