#!/bin/bash
# Quick PMC look at ONE workload with a chosen library (dev tool, GPU box).
# usage: tools/pmc_quick.sh OUTDIR LIB_OR_EMPTY PROGRAM...   e.g.  tools/pmc_quick.sh gpurun_out/q1 "" tools/bench_mixed.py
OUT=$1; LIB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/$OUT"
[ -n "$LIB" ] && export LGAR_LIB="$ROOT/$LIB"
cd /tmp; export TMPDIR=/tmp
pass() {
  name=$1; shift
  rm -rf /tmp/pq_$name
  rocprofv3 --kernel-trace --output-format csv "$@" -d /tmp/pq_$name -o p -- python3 "$ROOT/$PROG" $ARGS > "$ROOT/$OUT/$name.log" 2>&1 || echo "pass $name failed" >> "$ROOT/$OUT/$name.log"
  find /tmp/pq_$name \( -name "*counter_collection.csv" -o -name "*kernel_stats.csv" \) -size -20M -exec cp {} "$ROOT/$OUT/$name.csv" \;
  rm -rf /tmp/pq_$name
}
PROG=$1; shift; ARGS="$@"
pass stats --stats
pass cyc --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
pass ins --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_IFETCH
pass lvl --pmc SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE
pass typ --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU
