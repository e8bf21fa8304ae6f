"""Resource usage of the forward kernels of one measurement configuration (dev tool; hipcc only, no GPU):
python tools/resource_usage.py [-D... flags]   e.g.  python tools/resource_usage.py -DLGAR_ONLY_MIXED"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgar_py_amd import build as B
tmp = tempfile.mkdtemp()
flags = [f for f in B.FLAGS if f != "-shared"] + ["-DLGAR_NL=3", "-DLGAR_MEASURE", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
src = os.environ.get("LGAR_SRC", "lgar_kernels_nl.hip")
p = subprocess.run(["hipcc"] + flags + ["-c", os.path.join(B.CSRC, src), "-o", os.path.join(tmp, "k.o")], capture_output=True, text=True)
if p.returncode:
    print(p.stderr[-3000:]); sys.exit(1)
for blk in re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]:
    name = blk.split()[0]
    if "forward_kernel" not in name and "tangent_kernel" not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    get = lambda k: (re.search(k + r": (\d+)", blk) or [None, "?"])[1]
    print("%-50s VGPRs %s AGPRs %s spilled VGPRs %s SGPR spills %s scratch %s B/lane LDS %s B waves/SIMD %s" % (
        dem.replace("void lgar::", "").split("(")[0], get("VGPRs"), get("AGPRs"), get("VGPRs Spill"), get("SGPRs Spill"),
        get(r"ScratchSize \[bytes/lane\]"), get(r"LDS Size \[bytes/block\]"), get(r"Occupancy \[waves/SIMD\]")))
