mkdir -p gpurun_out/r3x
python -m pytest tests/test_ctm_gpu.py tests/test_ctm_fullsize_gpu.py tests/test_brca_gpu.py tests/test_inference_gpu.py tests/test_ctm_batch_gpu.py -m gpu -q 2>&1 | tail -3
MMM_CTM_DENSE=1 python -m pytest tests/test_ctm_gpu.py tests/test_brca_gpu.py tests/test_inference_gpu.py tests/test_ctm_batch_gpu.py -m gpu -q 2>&1 | tail -3
for c in 4 5; do
  for ll in 1 0; do
    MMM_CTM_LL_DENSE=$ll python bench.py --config $c --no-cpu-baseline > gpurun_out/r3x/cfg${c}_ll${ll}.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3x/*.json')):
    r=json.load(open(f)); print(f, "ms/step %.4f (%.4f-%.4f)" % (r["ms_per_step"], r["ms_per_step_min"], r["ms_per_step_max"]), {k:round(v,1) for k,v in r["iteration"]["kernel_us"].items()}, r["mma_evaluation_counts_equal_for_all_documents"], r["elbo_rel_err_vs_oracle"])
PY
