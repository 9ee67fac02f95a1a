#!/bin/bash
# rocprofv3 counter passes of bench.py, one pass per counter set (SQ: <= 8 counters per pass; FETCH_SIZE and WRITE_SIZE never share a
# pass -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  The profiled program comes directly after `--` (no env / sh hop).
# PMC_PROG=tools/diag_solve_split.py profiles that script instead of bench.py.
# usage: tools/pmc_run.sh <outdir under gpurun_out> <tag> <passes: e.g. "A B C D"> -- <bench.py arguments>
set -u
out=$1; tag=$2; passes=$3; shift 4
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/$out"
cd /tmp && export TMPDIR=/tmp
declare -A SETS
SETS[A]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD"
SETS[B]="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"
SETS[C]="FETCH_SIZE"
SETS[D]="WRITE_SIZE"
SETS[E]="GRBM_GUI_ACTIVE"
for p in $passes; do
  rocprofv3 --pmc ${SETS[$p]} --output-format csv -d "$R/$out/${tag}_$p" -o "${tag}_$p" -- python3 "$R/${PMC_PROG:-bench.py}" "$@" > "$R/$out/${tag}_$p.json" 2> "$R/$out/${tag}_$p.err" || echo "pass $p failed"
  echo "pass $p done: $(ls $R/$out/${tag}_$p 2>/dev/null | head -3 | tr '\n' ' ')"
done
