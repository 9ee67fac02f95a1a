#!/bin/bash
# The CTM part of tools/refresh_evidence.sh alone (counter passes of configs 4 / 5, then the default line, which attaches them): for a change that
# touched csrc/ctm* only.  usage (GPU box, repository root): bash tools/refresh_ctm_evidence.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; E=$R/gpurun_out/evidence; P=$E/pmc
mkdir -p $E; cd $R
bash tools/pmc_run.sh gpurun_out/evidence/pmc cfg4 "A B C D" -- --config 4 --no-cpu-baseline --repeats 3
bash tools/pmc_run.sh gpurun_out/evidence/pmc cfg5 "A B C D" -- --config 5 --no-cpu-baseline --repeats 3
python3 tools/pmc_summary.py $P cfg4 "k_ctm_solve_cpl<28" --json $E/traffic_ctm_solve_cfg4.json > $E/pmc_cfg4_solve.txt
python3 tools/pmc_summary.py $P cfg4 "k_ctm_theta_dense<10, 6>" > $E/pmc_cfg4_theta.txt
python3 tools/pmc_summary.py $P cfg4 k_ctm_loglik > $E/pmc_cfg4_loglik.txt
python3 tools/pmc_summary.py $P cfg4 k_ctm_moments > $E/pmc_cfg4_moments.txt
python3 tools/pmc_summary.py $P cfg5 k_ctm_solve_cpl --json $E/traffic_ctm_solve_cfg5.json > $E/pmc_cfg5_solve.txt
python3 tools/pmc_summary.py $P cfg5 "k_ctm_theta_dense<10, 6>" > $E/pmc_cfg5_theta.txt
python3 tools/pmc_summary.py $P cfg5 k_ctm_loglik > $E/pmc_cfg5_loglik.txt
rm -rf $P/*/
python3 tools/collect_evidence.py > /dev/null 2>&1
python3 bench.py > $E/bench_default.json 2> $E/bench_default.err; echo "default done"
ls $E | head -50
