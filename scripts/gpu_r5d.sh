#!/bin/bash
# Round 5, GPU call 4: per-launch timeline of ONE single-window encoder pass (which of the 51 GEMM launches are long?), Whisper-small and large-v3.
set -u
TAG=${1:-r5d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
cd /tmp
for m in small.en large-v3; do
  rm -rf /tmp/kt_$m; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_$m -o wlx --output-format csv -- python $REPO/scripts/encode_only.py $m 4 1 > "$OUT/enc_$m.log" 2>&1; echo "$m rc=$?"; tail -1 "$OUT/enc_$m.log"
  python - /tmp/kt_$m "$OUT/encoder_launches_$m.txt" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last encoder pass: from the last prep_window launch on
idx = max(i for i, r in enumerate(rows) if "prep_window" in r["Kernel_Name"])
seg = rows[idx:]
out = []
t_prev_end = None
tot = 0.0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - t_prev_end) / 1e3 if t_prev_end else 0.0
    t_prev_end = e
    nm = r["Kernel_Name"]
    nm = nm.replace("void wlx::", "").replace("(wlx::GemmParams)", "")[:60]
    g = [r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")]
    wg = [r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))]
    out.append("%-62s grid %-18s wg %-5s dur %8.2f us  gap %6.2f us" % (nm, "x".join(x for x in g if x), wg[0], (e - s) / 1e3, gap))
    tot += (e - s) / 1e3
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
out.append("launches %d  sum of durations %.1f us  span %.1f us" % (len(seg), tot, span))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:16] + ["..."] + out[-3:]))
PY
done
