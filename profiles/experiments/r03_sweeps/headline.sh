mkdir -p gpurun_out/r3h
timeout -k 10 600 python -m pytest tests/test_lda_gpu.py tests/test_ilda_gpu.py tests/test_brca_gpu.py -m gpu -x -q > gpurun_out/r3h/tests.txt 2>&1; tail -2 gpurun_out/r3h/tests.txt
for i in 1 2 3; do python bench.py --no-also --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r3h/cfg2_$i.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/cfg2_*.json')):
    r=json.load(open(f)); print(f, "us/step %.2f (%.2f-%.2f)" % (r["ms_per_step"]*1e3, r["ms_per_step_min"]*1e3, r["ms_per_step_max"]*1e3), {k:round(v,2) for k,v in r["iteration"]["kernel_us"].items()})
PY
