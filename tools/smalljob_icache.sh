# instruction-cache behaviour of a small job (64 columns x 64 cooperating lanes); dev tool, run on the GPU box
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
rm -rf /tmp/sp; rocprofv3 --kernel-trace --output-format csv --pmc $pass -d /tmp/sp -o p -- python3 $ROOT/tools/smalljob_run.py ${1:-64} ${2:-64} 2 > /tmp/sp.log 2>&1
grep "lanes" /tmp/sp.log | tail -1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/sp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lgar_forward_kernel' in r['Kernel_Name']: acc[(r['Kernel_Name'][:52], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, "max %.4g n=%d"%(max(v),len(v)))
PY
done
