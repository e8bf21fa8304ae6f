cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for L in 1 64; do
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64"; do
rm -rf /tmp/sp; rocprofv3 --kernel-trace --output-format csv --pmc $pass -d /tmp/sp -o p -- python3 $ROOT/tools/smalljob_run.py 64 $L 2 > /tmp/sp.log 2>&1
grep "lanes" /tmp/sp.log | tail -1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/sp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lgar_forward_kernel' in r['Kernel_Name']: acc[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print("lanes $L", k, "max %.4g n=%d"%(max(v),len(v)))
PY
done
done
