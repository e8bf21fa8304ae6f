"""Summarise tools/pmc_quick.sh output: per kernel, the median of each counter over its launches. usage: pmc_quick_summary.py DIR [kernel-substring]"""
import collections, csv, glob, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "forward_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/*.csv"):
    rd = csv.DictReader(open(f))
    if "Counter_Name" not in (rd.fieldnames or []):
        for r in rd:
            if sub in r.get("Name", ""): print("stats", r["Name"][:60], "calls", r["Calls"], "avg_ms", float(r["AverageNs"]) * 1e-6)
        continue
    for r in rd:
        if sub in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    if med.get("SQ_WAVE_CYCLES", 0) < 1e8 and med.get("SQ_INSTS", 0) < 1e8 and med.get("GRBM_GUI_ACTIVE", 0) < 1e7 and med.get("SQ_INSTS_VALU_TRANS_F32", 0) < 1e7: continue
    print(k)
    for n in sorted(med): print("   %-28s %.4g" % (n, med[n]))
