mkdir -p gpurun_out/r3x
timeout -k 10 900 python -m pytest tests/test_lda_gpu.py tests/test_lda_wide_gpu.py tests/test_ilda_gpu.py tests/test_brca_gpu.py tests/test_inference_gpu.py -m gpu -x -q > gpurun_out/r3x/lda_tests.txt 2>&1; tail -3 gpurun_out/r3x/lda_tests.txt
for d in 160000 640000; do
  python bench.py --config 2 --docs $d --no-cpu-baseline --no-also --steps 50 --warmup 5 --repeats 5 > gpurun_out/r3x/lda_${d}.json 2>gpurun_out/r3x/lda_${d}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3x/lda_*.json')):
    r=json.load(open(f)); print(f, "ms/step %.4f" % r["ms_per_step"], r["iteration"]["kernel_us"], r.get("elbo_rel_err_vs_oracle"), r.get("parity"))
PY
