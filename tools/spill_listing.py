"""Where does a kernel's scratch traffic come from?  (dev tool; hipcc only, no GPU)

python tools/spill_listing.py [KERNEL-SUBSTRING] [-D... flags]
    e.g.  python tools/spill_listing.py "Li8ELi3" -DLGAR_ONLY_MIXED

Compiles lgar_kernels_nl.hip (3 layers) to assembly with line tables and lists every scratch store / load of the chosen kernel
(default: the 8-slot mixed-precision forward kernel) with the source line it belongs to and its position in the instruction
stream.  Stores that sit inside the step loop are the ones that become HBM write traffic on every step (the prologue's are
executed once per column).  How round 5 found the GIUH queue, the LayerK struct passed by reference, the percolation sum, the
hand-over step and the atomics' zero offset (profiles/EXPERIMENTS.md, "Scratch write-back")."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgar_py_amd import build as B

args = sys.argv[1:]
kernel = "lgar_forward_kernelIdLi3ELi8ELi3"
if args and not args[0].startswith("-"):
    kernel = args.pop(0)
out = os.path.join(tempfile.mkdtemp(), "k.s")
flags = [f for f in B.FLAGS if f != "-shared"] + ["-DLGAR_NL=3", "-S", "--cuda-device-only", "-gline-tables-only"] + args
p = subprocess.run(["hipcc"] + flags + [os.path.join(B.CSRC, "lgar_kernels_nl.hip"), "-o", out], capture_output=True, text=True)
if p.returncode:
    print(p.stderr[-3000:])
    sys.exit(1)
txt = open(out).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
    files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
heads = [m for m in re.finditer(r"^(_Z\w+):", txt, re.M) if kernel in m.group(1)]
if not heads:
    print("no kernel matching %r" % kernel)
    sys.exit(1)
for h in heads:
    body = txt[h.start():txt.index(".Lfunc_end", h.start())].splitlines()
    name = subprocess.run(["c++filt", h.group(1)], capture_output=True, text=True).stdout.strip()
    print("== %s: %d lines of assembly" % (name, len(body)))
    loc = None
    for k, line in enumerate(body):
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            loc = "%s:%s" % (files.get(m.group(1), m.group(1)), m.group(2))
        if "scratch_store" in line or "scratch_load" in line:
            print("%-6s %6d  %-28s %s" % ("STORE" if "scratch_store" in line else "  load", k, loc, line.strip()[:80]))
