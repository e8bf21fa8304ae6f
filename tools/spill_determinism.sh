#!/bin/bash
# Does the register allocator's spill code change results from run to run?  (dev tool; build on the CPU container, run on the GPU box)
# usage: tools/spill_determinism.sh build | tools/spill_determinism.sh run OUTDIR
# Variants of the fp32 forward kernels with K values of register ballast per lane (csrc/lgar_measure.hpp LGAR_BALLAST): K more
# registers live through the whole time loop, i.e. K more spilled around the trapezoid at the kernel's 128-register budget; one of
# them with SGPR spills sent to scratch memory instead of VGPR lanes; and round 4's failing idea re-created (heads of the fp32
# trapezoid from the fronts' psi), plain, with the wave's LDS poisoned before every block, without SGPR->VGPR spills, at two waves per SIMD.  Each goes through tools/determinism_probe.py f32 (the
# 1 048 576-column job three times in one process: status words, series and front tables must be bit-identical).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
VARIANTS="nd_ballast8 nd_ballast16 nd_ballast24 nd_ballast16_nos2v nd_ballast24_O3 nd_heads nd_heads_poison nd_heads_nos2v nd_heads_occ2"
if [ "$1" = "build" ]; then
  python - <<'PY'
import sys; sys.path.insert(0, ".")
from lgar_py_amd import build as B
V = {"nd_ballast8": ["-DLGAR_ONLY_F32", "-DLGAR_BALLAST=8"], "nd_ballast16": ["-DLGAR_ONLY_F32", "-DLGAR_BALLAST=16"],
     "nd_ballast24": ["-DLGAR_ONLY_F32", "-DLGAR_BALLAST=24"],
     "nd_ballast16_nos2v": ["-DLGAR_ONLY_F32", "-DLGAR_BALLAST=16", "-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"],
     "nd_ballast24_O3": ["-DLGAR_ONLY_F32", "-DLGAR_BALLAST=24", "-O3"],
     # round 4's failing idea re-created: the fp32 calc_dzdt with the trapezoid's heads from the fronts' psi (csrc/lgar_measure.hpp)
     "nd_heads": ["-DLGAR_ONLY_F32", "-DLGAR_F32_HEADS_FROM_PSI"],
     "nd_heads_poison": ["-DLGAR_ONLY_F32", "-DLGAR_F32_HEADS_FROM_PSI", "-DLGAR_POISON_LDS"],
     "nd_heads_nos2v": ["-DLGAR_ONLY_F32", "-DLGAR_F32_HEADS_FROM_PSI", "-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"],
     "nd_heads_occ2": ["-DLGAR_ONLY_F32", "-DLGAR_F32_HEADS_FROM_PSI", "-DLGAR_OCC_F32_SMALL=2"]}
for k, v in V.items():
    print(k, B.build_variant(k, v), flush=True)
PY
  exit $?
fi
OUT=${2:-gpurun_out/spill_determinism}
mkdir -p "$OUT"
for v in $VARIANTS; do
  echo "== $v" >> "$OUT/det.log"
  LGAR_LIB=$ROOT/lgar_py_amd/csrc/variants/liblgar_hip_$v.so timeout -k 10 200 python tools/determinism_probe.py f32 >> "$OUT/det.log" 2>&1
done
grep -v amdgpu.ids "$OUT/det.log"
