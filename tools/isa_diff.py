"""Which kernels did a change touch?  Compiles lgar_kernels_nl.hip (3 soil layers, the product flags) from the working tree and
from a git revision, disassembles both code objects and compares every kernel instruction by instruction (branch targets and
literal addresses aside).  A kernel reported SAME executes the very instruction stream it did at that revision: its parity
sweeps, counters and timings carry over.  Runs where hipcc is (no GPU needed).  usage: python tools/isa_diff.py [REV=HEAD] [UNIT]
(dev tool)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgar_py_amd import build as B

LLVM = "/opt/rocm/lib/llvm/bin"
rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
unit = sys.argv[2] if len(sys.argv) > 2 else "lgar_kernels_nl.hip"
flags = [f for f in B.FLAGS if f != "-shared"] + ["-DLGAR_NL=3"]


def kernels(src_root, tag, tmp):
    obj = os.path.join(tmp, tag + ".o")
    subprocess.check_call(["hipcc"] + flags + ["-c", os.path.join(src_root, "lgar_py_amd", "csrc", unit), "-o", obj])
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + obj + ".fat", obj])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + obj + ".fat", "--output=" + obj + ".co"])
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", obj + ".co"], capture_output=True, text=True).stdout
    res, cur = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            res[cur] = []
        elif cur and ln.strip():
            res[cur].append(re.sub(r"0x[0-9a-f]+|\d+$", "", ln.split("//")[0].strip()))
    return res


with tempfile.TemporaryDirectory() as tmp:
    old_root = os.path.join(tmp, "old")
    for rel in ["include/lgar.h"] + ["lgar_py_amd/csrc/" + f for f in os.listdir(os.path.join(ROOT, "lgar_py_amd", "csrc"))
                                      if f.endswith((".hpp", ".hip"))]:
        p = subprocess.run(["git", "-C", ROOT, "show", "%s:%s" % (rev, rel)], capture_output=True)
        if p.returncode == 0:
            os.makedirs(os.path.dirname(os.path.join(old_root, rel)), exist_ok=True)
            open(os.path.join(old_root, rel), "wb").write(p.stdout)
    a, b = kernels(old_root, "old", tmp), kernels(ROOT, "new", tmp)
    for k in sorted(set(a) | set(b)):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("void lgar::", "").split("(")[0]
        print("%-5s %6d -> %6d instructions  %s" % ("SAME" if a.get(k) == b.get(k) else "DIFF", len(a.get(k, [])), len(b.get(k, [])), name))
