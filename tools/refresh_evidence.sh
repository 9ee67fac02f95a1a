#!/bin/bash
# Regenerates the measured evidence of profiles/ for the current build on the GPU box (one MI355X): bench lines, rocprofv3 kernel summaries,
# counter passes.  Everything lands under gpurun_out/evidence/; tools/collect_evidence.py then copies the summaries into profiles/.
# usage (on the GPU box, from the repository root): bash tools/refresh_evidence.sh [bench|pmc]
# optional argument: "bench" (bench lines + kernel summaries), "pmc" (counter passes + their summaries) or nothing (counters first, then the bench
# lines, which attach them; > 20 minutes)
set -u
STAGE=${1:-all}
R=${GRAFT_REPO_ROOT:-$PWD}
E=$R/gpurun_out/evidence
P=$E/pmc
mkdir -p $E
cd $R
if [ "$STAGE" != "bench" ]; then
echo "== counter passes"
cd $R
bash tools/pmc_run.sh gpurun_out/evidence/pmc lda10k "A B C D" -- --no-cpu-baseline --no-also --repeats 3
bash tools/pmc_run.sh gpurun_out/evidence/pmc lda160k "A B C D" -- --docs 160000 --no-cpu-baseline --repeats 3 --steps 20
bash tools/pmc_run.sh gpurun_out/evidence/pmc lda640k "A B C D" -- --docs 640000 --no-cpu-baseline --repeats 3 --steps 10
bash tools/pmc_run.sh gpurun_out/evidence/pmc cfg4 "A B C D" -- --config 4 --no-cpu-baseline --repeats 3
bash tools/pmc_run.sh gpurun_out/evidence/pmc cfg5 "A B C D" -- --config 5 --no-cpu-baseline --repeats 3
python3 tools/pmc_summary.py $P lda10k "k_lda_estep<" --json $E/traffic_lda_estep.json > $E/pmc_lda10k_estep.txt
python3 tools/pmc_summary.py $P lda10k k_lda_reduce_ll_mstep > $E/pmc_lda10k_merged.txt
python3 tools/pmc_summary.py $P lda160k k_lda_estep_dense > $E/pmc_lda160k_estep_dense.txt
python3 tools/pmc_summary.py $P lda640k k_lda_estep_dense --json $E/traffic_lda_estep_dense_640k.json > $E/pmc_lda640k_estep_dense.txt
python3 tools/pmc_summary.py $P lda640k k_lda_reduce_ll_mstep > $E/pmc_lda640k_merged.txt
python3 tools/pmc_summary.py $P cfg4 "k_ctm_solve_cpl<28" --json $E/traffic_ctm_solve_cfg4.json > $E/pmc_cfg4_solve.txt
python3 tools/pmc_summary.py $P cfg4 "k_ctm_theta_dense<10, 6>" > $E/pmc_cfg4_theta.txt
python3 tools/pmc_summary.py $P cfg4 k_ctm_loglik > $E/pmc_cfg4_loglik.txt
python3 tools/pmc_summary.py $P cfg4 k_ctm_moments > $E/pmc_cfg4_moments.txt
python3 tools/pmc_summary.py $P cfg5 k_ctm_solve_cpl --json $E/traffic_ctm_solve_cfg5.json > $E/pmc_cfg5_solve.txt
python3 tools/pmc_summary.py $P cfg5 "k_ctm_theta_dense<10, 6>" > $E/pmc_cfg5_theta.txt
python3 tools/pmc_summary.py $P cfg5 k_ctm_loglik > $E/pmc_cfg5_loglik.txt
# the bench lines below attach a counter file only if it sits under profiles/ and its hash matches the sources: copy the fresh summaries there first
rm -rf $P/*/
python3 tools/collect_evidence.py > /dev/null 2>&1
fi
if [ "$STAGE" != "pmc" ]; then
echo "== bench lines"
python3 bench.py > $E/bench_default.json 2> $E/bench_default.err; echo "default (cfg 2 + also cfg 4, 5) done"
for c in 4 5; do python3 bench.py --config $c > $E/bench_cfg$c.json 2> $E/bench_cfg$c.err; echo "cfg $c done"; done
python3 bench.py --gpus 2 --no-also --no-cpu-baseline > $E/bench_2ranks_one_card.json 2> $E/bench_2ranks_one_card.err; echo "2 self-launched ranks done"
python3 bench.py --gpus 2 > $E/bench_2ranks_one_card_default.json 2> $E/bench_2ranks_one_card_default.err; echo "2 self-launched ranks, default line done"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $E/bench_cfg2_steps20.json 2> /dev/null
: > $E/lda_scaling.jsonl
for D in 10000 40000 160000 640000; do python3 bench.py --docs $D --no-cpu-baseline >> $E/lda_scaling.jsonl 2> /dev/null; done
python3 bench.py --docs 640000 --no-cpu-baseline --lda-build sparse > $E/lda_640k_csr.json 2> /dev/null
echo "== shard sizes (what one GPU of an N-GPU strong run holds) and the solve layouts at those sizes"
python3 tools/shard_sizes.py > $E/shard_sizes.jsonl 2> $E/shard_sizes.err; echo "shard sizes done"
python3 tools/solve_layouts.py > $E/solve_layouts.jsonl 2> $E/solve_layouts.err; echo "solve layouts done"
echo "== kernel summaries"
cd /tmp && export TMPDIR=/tmp
for c in 2 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $E/ks_cfg$c -o cfg$c -- python3 $R/bench.py --config $c --no-cpu-baseline --no-also --repeats 3 > $E/ks_cfg$c.json 2> $E/ks_cfg$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $E/ks_lda640k -o lda640k -- python3 $R/bench.py --docs 640000 --no-cpu-baseline --repeats 3 > $E/ks_lda640k.json 2> $E/ks_lda640k.err
fi
# the raw counter tables are large: only the summaries travel back
rm -rf $P/*/ $E/ks_*/*_kernel_trace.csv $E/ks_*/*agent_info.csv $E/ks_*/*domain_stats.csv
ls $E
