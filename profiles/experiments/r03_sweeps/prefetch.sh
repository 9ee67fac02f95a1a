mkdir -p gpurun_out/r3y
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3y/tests.txt 2>&1; tail -3 gpurun_out/r3y/tests.txt
for c in 4 5; do
  python bench.py --config $c --no-cpu-baseline --repeats 5 > gpurun_out/r3y/cfg${c}.json 2>gpurun_out/r3y/cfg${c}.err
done
python bench.py --config 2 --docs 640000 --no-cpu-baseline --no-also --steps 50 --warmup 5 --repeats 5 > gpurun_out/r3y/lda_640000.json 2>gpurun_out/r3y/lda_640000.err
python bench.py --no-also --no-cpu-baseline > gpurun_out/r3y/cfg2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3y/*.json')):
    r=json.load(open(f)); print(f, "ms/step %.4f" % r["ms_per_step"], {k:round(v,1) for k,v in r["iteration"]["kernel_us"].items()}, r.get("mma_evaluation_counts_equal_for_all_documents"), r.get("elbo_rel_err_vs_oracle"))
PY
