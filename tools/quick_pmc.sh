cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for c in WRITE_SIZE FETCH_SIZE SQ_INSTS_VMEM; do
rm -rf /tmp/qp; rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/qp -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/qp.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob('/tmp/qp/**/*counter_collection.csv', recursive=True):
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'lgar_forward_kernel<float, 3, 8, 1>' in r['Kernel_Name'] and r['Counter_Name']=='$c']
    v.sort(); print('$c', v[len(v)//2] if v else None)
PY
done
grep '^{' /tmp/qp.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
