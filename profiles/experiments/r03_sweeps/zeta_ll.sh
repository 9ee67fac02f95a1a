mkdir -p gpurun_out/r3q
timeout -k 10 1100 python -m pytest tests/test_ctm_gpu.py tests/test_brca_gpu.py tests/test_ctm_batch_gpu.py tests/test_ctm_fullsize_gpu.py tests/test_ctm_wide_gpu.py tests/test_inference_gpu.py tests/test_c_example.py -m gpu -x -q > gpurun_out/r3q/tests.txt 2>&1; tail -3 gpurun_out/r3q/tests.txt
for c in 4 5; do
  for o in 0 1; do
    MMM_CTM_ZETA_LL=$o python bench.py --config $c --no-cpu-baseline --repeats 5 > gpurun_out/r3q/cfg${c}_z${o}.json 2>gpurun_out/r3q/cfg${c}_z${o}.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3q/*.json')):
    r=json.load(open(f)); print(f, "ms/step %.4f (%.4f-%.4f)" % (r["ms_per_step"], r["ms_per_step_min"], r["ms_per_step_max"]), {k:round(v,1) for k,v in r["iteration"]["kernel_us"].items()}, r.get("mma_evaluation_counts_equal_for_all_documents"), r.get("elbo_rel_err_vs_oracle"))
PY
