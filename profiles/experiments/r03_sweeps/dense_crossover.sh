mkdir -p gpurun_out/r3z
for d in 10000 15000 20000 25000 30000 40000; do
  for m in 0 1; do
    MMM_LDA_DENSE=$m python bench.py --config 2 --docs $d --no-cpu-baseline --no-also --steps 50 --warmup 5 --repeats 5 > gpurun_out/r3z/lda_${d}_dense${m}.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3z/*.json')):
    r=json.load(open(f)); print(f, "ms/step %.4f" % r["ms_per_step"], {k:round(v,1) for k,v in r["iteration"]["kernel_us"].items()})
PY
