mkdir -p gpurun_out/r3i
python -m pytest tests/test_lda_gpu.py -m gpu -x -q > gpurun_out/r3i/lda_tests.log 2>&1; tail -3 gpurun_out/r3i/lda_tests.log
for D in 160000 640000; do
  python bench.py --docs $D --no-cpu-baseline --repeats 5 --steps 20 > gpurun_out/r3i/d32_$D.json 2> gpurun_out/r3i/d32_$D.err
  MMM_LDA_DENSE32=0 python bench.py --docs $D --no-cpu-baseline --repeats 5 --steps 20 > gpurun_out/r3i/d16_$D.json 2> gpurun_out/r3i/d16_$D.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/d*.json')):
    try:
        r=json.load(open(f)); print(f, "ms/step %.4f" % r["ms_per_step"], r["iteration"]["kernel_us"], r["roofline"]["kernel"], "%.0f GB/s" % r["roofline"]["achieved"], r["elbo_rel_err_vs_oracle"])
    except Exception as e: print(f, "ERR", e)
PY
