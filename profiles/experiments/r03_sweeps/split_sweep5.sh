mkdir -p gpurun_out/r3h
for s in 2:3:2:2 2:2:2:2; do
MMM_CTM_SPLIT=$s timeout -k 10 120 python tools/diag_solve_split.py 5 > gpurun_out/r3h/split_dyn_$s.json 2> gpurun_out/r3h/err_$s || echo FAILED $s
MMM_CTM_CLAIM=0 MMM_CTM_SPLIT=$s timeout -k 10 120 python tools/diag_solve_split.py 5 > gpurun_out/r3h/split_static_$s.json 2> gpurun_out/r3h/err2_$s || echo FAILED $s
done
timeout -k 10 120 python tools/diag_solve_split.py 5 > gpurun_out/r3h/split_default.json 2> gpurun_out/r3h/err_d || echo FAILED
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/split_*.json')):
    try:
        r=json.load(open(f)); print(f.split('split_')[1][:-5], "fused %.0f nu %.0f lam %.0f" % (r["fused_solve_us"], r["nu_us"], r["lambda_us"]), r["evals_per_doc"], r["stage_evals_per_doc"])
    except Exception as e: print(f, "ERR", e)
PY
