mkdir -p gpurun_out/r3d
MMM_CTM_CPL=4 python tools/diag_solve_split.py 4 > gpurun_out/r3d/split_cpl4.json 2> gpurun_out/r3d/err_cpl4 || echo FAILED
MMM_CTM_CPL=4 MMM_CTM_GRID_SOLVE=1024 python tools/diag_solve_split.py 4 > gpurun_out/r3d/split_cpl4_g1024.json 2> gpurun_out/r3d/err_cpl4b || echo FAILED
MMM_CTM_SPLIT=16:3:16:3 python tools/diag_solve_split.py 4 > gpurun_out/r3d/split_16316.json 2> gpurun_out/r3d/err_s || echo FAILED
python tools/diag_solve_split.py 4 > gpurun_out/r3d/split_default.json
MMM_CTM_CPL=4 python tools/diag_solve_split.py 4 50000 60 > gpurun_out/r3d/split_cpl4_p60.json 2> gpurun_out/r3d/err_cpl4c || echo FAILED
python tools/diag_solve_split.py 4 50000 60 > gpurun_out/r3d/split_default_p60.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/split_*.json')):
    try:
        r=json.load(open(f)); print(f.split('split_')[1][:-5], "fused %.0f nu %.0f lam %.0f" % (r["fused_solve_us"], r["nu_us"], r["lambda_us"]), r["evals_per_doc"], r["stage_evals_per_doc"])
    except Exception as e: print(f, "ERR", e)
PY
