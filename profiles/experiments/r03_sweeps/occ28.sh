mkdir -p gpurun_out/r3r
for occ in 3 4; do
  MMM_CTM_OCC28=$occ python tools/diag_solve_split.py 4 50000 20 > gpurun_out/r3r/occ${occ}_p20.json 2>/dev/null
  MMM_CTM_OCC28=$occ python tools/diag_solve_split.py 4 50000 60 > gpurun_out/r3r/occ${occ}_p60.json 2>/dev/null
  MMM_CTM_OCC28=$occ python bench.py --config 4 --no-cpu-baseline > gpurun_out/r3r/bench_occ${occ}.json 2>/dev/null
done
python tools/diag_solve_split.py 5 100000 20 > gpurun_out/r3r/cfg5_p20.json 2>/dev/null
python bench.py --config 5 --no-cpu-baseline > gpurun_out/r3r/bench_cfg5.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3r/*.json')):
    r=json.load(open(f))
    if "fused_solve_us" in r: print(f, "fused %.0f nu %.0f lam %.0f" % (r["fused_solve_us"], r["nu_us"], r["lambda_us"]), r["evals_per_doc"])
    else: print(f, "ms/step %.4f" % r["ms_per_step"], r["iteration"]["kernel_us"], r["mma_evaluation_counts_equal_for_all_documents"], r["elbo_rel_err_vs_oracle"])
PY
