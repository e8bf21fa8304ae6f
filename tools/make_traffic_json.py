"""profiles/<round>/traffic.json from the rocprofv3 passes of tools/pmc_passes.sh (dev tool).
usage: make_traffic_json.py PROF_DIR OUT_JSON [kernel-substring] [columns] [timesteps] [dtype] [csv-prefix]
PROF_DIR/library_fingerprint (written by pmc_passes.sh on the GPU box) records which library build the counters belong to."""
import collections
import csv
import glob
import json
import sys

prof, out = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else "lgar_forward_kernel<float, 3, 8, 1>"
cols = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
T = int(sys.argv[5]) if len(sys.argv) > 5 else 144
dtype = sys.argv[6] if len(sys.argv) > 6 else "f32"
prefix = sys.argv[7] if len(sys.argv) > 7 else {"f32": "f32_", "f64": "f64_"}[dtype]
import os
fp = None
if os.path.exists(os.path.join(prof, "library_fingerprint")):
    fp = open(os.path.join(prof, "library_fingerprint")).read().strip()
acc = collections.defaultdict(list)
for f in glob.glob(prof + "/" + prefix + "*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
ms = None
for f in glob.glob(prof + "/" + prefix + "stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Name"]:
            ms = float(r["AverageNs"]) * 1e-6
steps = (cols / 64) * T  # wave-steps: 64-column blocks x forcing steps (the persistent grid launches fewer waves than blocks)
valu = {
    "bound": "valu-issue", "kernel": kern, "kernel_ms_rocprof": ms,
    "SQ_INSTS_VALU": med.get("SQ_INSTS_VALU"), "SQ_INSTS_VALU_TRANS_F32": med.get("SQ_INSTS_VALU_TRANS_F32"),
    "SQ_INSTS_SALU": med.get("SQ_INSTS_SALU"), "SQ_INSTS_LDS": med.get("SQ_INSTS_LDS"), "SQ_INSTS_VMEM": med.get("SQ_INSTS_VMEM"),
    "SQ_ACTIVE_INST_VALU_quadcycles": med.get("SQ_ACTIVE_INST_VALU"), "SQ_WAVE_CYCLES": med.get("SQ_WAVE_CYCLES"),
    "SQ_WAIT_INST_ANY": med.get("SQ_WAIT_INST_ANY"), "GRBM_GUI_ACTIVE": med.get("GRBM_GUI_ACTIVE"),
    "valu_insts_per_wave_step": (med.get("SQ_INSTS_VALU", 0) / steps) if steps else None,
    "transcendentals_per_wave_step": (med.get("SQ_INSTS_VALU_TRANS_F32", 0) / steps) if steps else None,
    "salu_per_valu": (med.get("SQ_INSTS_SALU", 0) / med["SQ_INSTS_VALU"]) if med.get("SQ_INSTS_VALU") else None,
    "active_lane_fraction": (med["SQ_THREAD_CYCLES_VALU"] / (64 * med["SQ_ACTIVE_INST_VALU"])) if med.get("SQ_ACTIVE_INST_VALU") else None,
    "busy_frac": (4 * med["SQ_ACTIVE_INST_VALU"] / (1024 * med["GRBM_GUI_ACTIVE"] / 8)) if med.get("GRBM_GUI_ACTIVE") and med.get("SQ_ACTIVE_INST_VALU") else None,
    "definition": "medians over the launches of separate rocprofv3 --pmc passes (tools/pmc_passes.sh); busy_frac = 4*SQ_ACTIVE_INST_VALU "
                  "/ (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs); per wave-step = per 64 columns x 1 forcing step",
}
rec = {"columns": cols, "timesteps": T, "dtype": dtype, "library_fingerprint": fp, "FETCH_SIZE_KB": med.get("FETCH_SIZE"), "WRITE_SIZE_KB": med.get("WRITE_SIZE"),
       "note": "FETCH_SIZE is doubled by bench.py as MI355X_MICROARCH.md prescribes for gfx950", "valu": valu}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
