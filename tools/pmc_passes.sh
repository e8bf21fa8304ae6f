#!/bin/bash
# rocprofv3 --pmc passes over a short bench.py run (dev tool, GPU box).  usage: tools/pmc_passes.sh OUTDIR [bench args...]
# Counters are collected in separate passes (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share one), with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes.  Summaries: tools/pmc_summary.py OUTDIR/*
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/pmc_$name -o p -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/$name.log" 2>&1 || echo "pass $name failed" >> "$ROOT/$OUT/$name.log"
  mkdir -p "$ROOT/$OUT/$name"
  find /tmp/pmc_$name -name "*counter_collection.csv" -size -20M -exec cp {} "$ROOT/$OUT/$name/" \;
  rm -rf /tmp/pmc_$name
}
pass valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass mix SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
