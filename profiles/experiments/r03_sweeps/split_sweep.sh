mkdir -p gpurun_out/r3c
for s in 8:3:16:3 16:4:16:3 16:4:16:4 8:3:32:4 16:4:32:4 8:2:16:3 16:3:16:3; do
  MMM_CTM_SPLIT=$s python tools/diag_solve_split.py 4 > gpurun_out/r3c/split_$s.json 2> gpurun_out/r3c/err_$s || echo "FAILED $s"
done
python tools/diag_solve_split.py 4 > gpurun_out/r3c/split_default.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c/split_*.json')):
    try:
        r=json.load(open(f)); print(f.split('split_')[1][:-5], "fused %.0f nu %.0f lam %.0f" % (r["fused_solve_us"], r["nu_us"], r["lambda_us"]), r["evals_per_doc"], r["stage_evals_per_doc"])
    except Exception as e: print(f, "ERR", e)
PY
