mkdir -p gpurun_out/r3u
for c in 4 5; do
  for dmode in 1 0; do
    MMM_CTM_DENSE=$dmode python bench.py --config $c --no-cpu-baseline > gpurun_out/r3u/cfg${c}_dense${dmode}.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3u/*.json')):
    r=json.load(open(f)); print(f, "ms/step %.4f (%.4f-%.4f)" % (r["ms_per_step"], r["ms_per_step_min"], r["ms_per_step_max"]), {k:round(v,1) for k,v in r["iteration"]["kernel_us"].items()}, r["mma_evaluation_counts_equal_for_all_documents"], r["elbo_rel_err_vs_oracle"], r["theta_max_rel_err_vs_oracle"])
PY
