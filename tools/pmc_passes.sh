#!/bin/bash
# rocprofv3 passes over short runs (dev tool, GPU box).  usage: tools/pmc_passes.sh OUTDIR
#   f32:  bench.py (1M columns fp32, the timed configuration)      f64: bench.py --dtype f64
#   tan:  tools/bench_autograd.py 100000 f64 (BASELINE configs[4]: forward + the tangent launch of the 9 parameter directions)
#   mix:  tools/bench_mixed.py (1M columns, fp64 state with the fp32-transcendental trapezoid)
# Counters are collected in separate passes (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share one), with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes; a plain --kernel-trace --stats pass gives the kernel times.
# Only the small CSVs are copied into OUTDIR (the raw rocprofv3 output stays in /tmp on the box).
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/$OUT"
# which library build these counters belong to (bench.py reports roofline.traffic only for a matching build)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from lgar_py_amd import build as B; print(B._fingerprint())" > "$ROOT/$OUT/library_fingerprint"
cd /tmp; export TMPDIR=/tmp
run() {  # run NAME WORKLOAD rocprof-args...
  name=$1; wl=$2; shift 2
  case $wl in
    f32) PROG="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras" ;;
    f64) PROG="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --dtype f64" ;;
    tan) PROG="python3 $ROOT/tools/bench_autograd.py 100000 f64" ;;
    mix) PROG="python3 $ROOT/tools/bench_mixed.py" ;;
  esac
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --output-format csv "$@" -d /tmp/prof_$name -o p -- $PROG > "$ROOT/$OUT/${wl}_$name.log" 2>&1 || echo "pass $name failed" >> "$ROOT/$OUT/${wl}_$name.log"
  mkdir -p "$ROOT/$OUT/${wl}_$name"
  find /tmp/prof_$name \( -name "*counter_collection.csv" -o -name "*kernel_stats.csv" \) -size -20M -exec cp {} "$ROOT/$OUT/${wl}_$name/" \;
  rm -rf /tmp/prof_$name
}
for wl in ${WORKLOADS:-f32 f64 tan mix}; do
  echo "workload $wl: $(date +%T)"
  run stats $wl --stats
  run valu $wl --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run mix $wl --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64
  run wait $wl --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
  run fetch $wl --pmc FETCH_SIZE GRBM_GUI_ACTIVE
  run write $wl --pmc WRITE_SIZE
done
