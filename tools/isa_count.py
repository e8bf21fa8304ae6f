"""ISA-level instruction census of the compiled forward kernels (dev tool; runs where hipcc is, no GPU needed).
Compiles lgar_kernels_nl.hip for 3 layers with -save-temps, prints resource usage per forward kernel, the static
instruction histogram of the fp32 fast kernel, and the body of its hot loop (the 4-node Geff iteration) with a count of
transcendental / packed / plain vector instructions.  usage: python tools/isa_count.py [OUT.txt]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgar_py_amd import build as B

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
tmp = tempfile.mkdtemp()
flags = [f for f in B.FLAGS if f != "-shared"] + ["-DLGAR_NL=3", "-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
p = subprocess.run(["hipcc"] + flags + ["-c", os.path.join(B.CSRC, "lgar_kernels_nl.hip"), "-o", os.path.join(tmp, "k.o")],
                   capture_output=True, text=True)
rem = p.stderr
print("== resource usage (hipcc -Rpass-analysis=kernel-resource-usage), 3 soil layers ==", file=out)
for blk in re.split(r"remark: [^\n]*Function Name: ", rem)[1:]:
    name = blk.split()[0]
    if "forward_kernel" not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    get = lambda k: (re.search(k + r": (\d+)", blk) or [None, "?"])[1]
    print("%-62s VGPRs %s  spilled VGPRs %s  SGPR spills %s  scratch %s B/lane  LDS %s B  waves/SIMD %s" % (
        dem.replace("void lgar::", "").split("(")[0], get("VGPRs"), get("VGPRs Spill"), get("SGPRs Spill"),
        get(r"ScratchSize \[bytes/lane\]"), get(r"LDS Size \[bytes/block\]"), get(r"Occupancy \[waves/SIMD\]")), file=out)
asm = open(os.path.join(tmp, "lgar_kernels_nl-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
name = "_ZN4lgar19lgar_forward_kernelIfLi3ELi8ELi1EEEvNS_5KArgsIT_EE"
i = asm.index(name + ":")
body = asm[i:asm.index("s_endpgm", i)]
ops = collections.Counter()
for ln in body.splitlines():
    t = ln.strip()
    if t and not t.startswith((";", ".")) and not t.endswith(":"):
        ops[t.split()[0]] += 1
print("\n== lgar_forward_kernel<float, 3, 8, 1>: %d static instructions; most frequent ==" % sum(ops.values()), file=out)
print(", ".join("%s %d" % kv for kv in ops.most_common(24)), file=out)


def cat(op):
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
        return "transcendental (8 cycles)"
    if op.startswith("v_pk_"):
        return "packed fp32 (4 cycles, two lanes' worth)"
    if op.startswith("v_"):
        return "plain vector"
    if op.startswith("s_"):
        return "scalar"
    return "memory/other"


blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
for k in range(1, len(blocks), 2):
    blk = blocks[k + 1].split("\n.LBB")[0]
    ins = [l.strip() for l in blk.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    c = collections.Counter(x.split()[0] for x in ins)
    if c.get("v_log_f32_e32", 0) == 8 and c.get("v_exp_f32_e32", 0) == 8:
        print("\n== hot loop: one iteration = FOUR trapezoid nodes of calc_geff (lgar/green_ampt.py:66-84) ==", file=out)
        cc = collections.Counter(cat(x.split()[0]) for x in ins)
        for kk, v in cc.items():
            print("  %-44s %3d per iteration = %.2f per node" % (kk, v, v / 4.0), file=out)
        print("  issue cycles per node at the measured costs (tools/valu_probe.py): %.1f (transcendentals %.0f)" % (
            (8 * cc["transcendental (8 cycles)"] + 4 * cc["packed fp32 (4 cycles, two lanes' worth)"] + 4 * cc.get("plain vector", 0)) / 4.0,
            8 * cc["transcendental (8 cycles)"] / 4.0), file=out)
        print("\n".join("    " + x for x in ins), file=out)
        break
