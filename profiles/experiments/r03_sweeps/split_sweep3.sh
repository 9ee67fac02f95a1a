mkdir -p gpurun_out/r3e
for s in 16:3:16:3 16:4:16:3 16:3:32:4; do
MMM_CTM_SPLIT=$s timeout -k 10 120 python tools/diag_solve_split.py 4 > gpurun_out/r3e/split_dyn_$s.json 2> gpurun_out/r3e/err_$s || echo FAILED $s
MMM_CTM_CLAIM=0 MMM_CTM_SPLIT=$s timeout -k 10 120 python tools/diag_solve_split.py 4 > gpurun_out/r3e/split_static_$s.json 2> gpurun_out/r3e/err2_$s || echo FAILED $s
done
MMM_CTM_SPLIT=16:3:16:3 timeout -k 10 120 python tools/diag_solve_split.py 4 50000 60 > gpurun_out/r3e/split_dyn_p60.json 2> gpurun_out/r3e/err_p60 || echo FAILED
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/split_*.json')):
    try:
        r=json.load(open(f)); print(f.split('split_')[1][:-5], "fused %.0f nu %.0f lam %.0f" % (r["fused_solve_us"], r["nu_us"], r["lambda_us"]), r["evals_per_doc"], r["stage_evals_per_doc"])
    except Exception as e: print(f, "ERR", e)
PY
