mkdir -p gpurun_out/r3w
for c in 4 5; do
  for gr in 256 512 1024; do
    MMM_CTM_GRID=$gr python bench.py --config $c --no-cpu-baseline --repeats 5 > gpurun_out/r3w/cfg${c}_grid${gr}.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3w/*.json')):
    r=json.load(open(f)); print(f, "ms/step %.4f" % r["ms_per_step"], {k:round(v,1) for k,v in r["iteration"]["kernel_us"].items()}, r["mma_evaluation_counts_equal_for_all_documents"])
PY
