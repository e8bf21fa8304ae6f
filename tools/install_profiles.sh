#!/bin/bash
# Copy the small CSVs of a tools/pmc_passes.sh run into profiles/<round>/ under the committed names and regenerate the
# derived files (dev tool, runs in the build container).  usage: tools/install_profiles.sh gpurun_out/<prof dir> profiles/r02
SRC=$1; DST=$2
mkdir -p "$DST/pmc"
for wl in f32 f64 tan mix; do
  [ -f "$SRC/${wl}_stats/p_kernel_stats.csv" ] && cp "$SRC/${wl}_stats/p_kernel_stats.csv" "$DST/${wl}_kernel_stats.csv"
  for pass in valu mix wait fetch write; do
    [ -f "$SRC/${wl}_$pass/p_counter_collection.csv" ] && cp "$SRC/${wl}_$pass/p_counter_collection.csv" "$DST/pmc/${wl}_${pass}_counter_collection.csv"
  done
done
# keep only the rows of this library's kernels (a bench run launches thousands of small torch kernels too)
for f in "$DST"/pmc/*_counter_collection.csv; do
  python - "$f" <<'PY'
import sys
f = sys.argv[1]
rows = open(f).read().splitlines()
keep = [rows[0]] + [r for r in rows[1:] if "lgar_" in r]
open(f, "w").write("\n".join(keep) + "\n")
PY
done
python tools/pmc_summary.py "$SRC" > "$DST/pmc_summary.txt" 2>/dev/null
python tools/make_traffic_json.py "$SRC" "$DST/traffic.json" "lgar_forward_kernel<float, 3, 8, 1>" 1048576 144 f32
python tools/make_traffic_json.py "$SRC" "$DST/traffic_f64.json" "lgar_forward_kernel<double, 3, 8, 1>" 1048576 144 f64
python tools/make_traffic_json.py "$SRC" "$DST/traffic_mixed.json" "lgar_forward_kernel<double, 3, 8, 3>" 1048576 144 f64mix mix_
python tools/make_traffic_json.py "$SRC" "$DST/traffic_tangent.json" "lgar_tangent_kernel<double, 3, 8, 1>" 900000 144 f64tan tan_
cp "$SRC/library_fingerprint" "$DST/library_fingerprint" 2>/dev/null
python tools/isa_count.py "$DST/isa_census.txt"
