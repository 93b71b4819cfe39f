#!/bin/bash
# Round-2 pass 4: XCD-aware GEMM tile map, 64x96 tiles, VGPR-form MFMA accumulators, trimmed attention VALU.
set -u
TAG=${1:-r2d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" "$OUT/pytest.log" | tail -3
grep -E "12-layer|32-layer|common prefix|^FAILED|^E  " "$OUT/pytest.log" | head -30
echo "--- encoder A/B (encode_ms per 30 s window, small.en, 6 encodes)"
for cfg in "X=0" "WLX_GEMM2_XCD=0" "WLX_ENC_ATTN=1" "WLX_ENC_GEMM=1" "WLX_GEMM2_SHAPE=0" "WLX_GEMM2_SHAPE=6" "WLX_GEMM2_SHAPE=5" "WLX_GEMM2_SHAPE=4" "WLX_GEMM2_SHAPE=1"; do
  echo -n "[$cfg] "; env $cfg timeout 120 python scripts/encode_only.py small.en 6 2>&1 | grep encode_ms
done
echo -n "[large-v3 new] "; timeout 300 python scripts/encode_only.py large-v3 3 2>&1 | grep encode_ms
echo -n "[large-v3 old] "; WLX_ENC_GEMM=1 WLX_ENC_ATTN=1 timeout 300 python scripts/encode_only.py large-v3 3 2>&1 | grep encode_ms
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof_enc" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 4 > "$OUT/rocprof_enc.log" 2>&1; echo "rocprof enc rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof_enc" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && head -12 "$F"
python - "$OUT/rocprof_enc" <<'PY'
import csv, glob, sys, collections
# per-GEMM-shape durations: group gemm2 launches by grid size
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "gemm2" in n or "attn" in n:
        key = (n[:40], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
        acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items()):
    v = sorted(v); print(k, len(v), "median us %.1f min %.1f" % (v[len(v)//2] / 1e3, v[0] / 1e3))
PY
find "$OUT" -name '*kernel_trace.csv' -size +5M -delete
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace -d "$OUT/pmc_sq" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 > "$OUT/pmc_sq.log" 2>&1; echo "pmc rc=$?"
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d "$OUT/pmc_mem" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 > "$OUT/pmc_mem.log" 2>&1
cd "$REPO"
for n in sq mem; do python scripts/pmc_summary.py "$OUT/pmc_$n" 2>/dev/null | grep -E "gemm2|attn_encoder" > "$OUT/pmc_${n}_summary.csv"; cut -c1-140 "$OUT/pmc_${n}_summary.csv"; find "$OUT/pmc_$n" -name '*.csv' -size +3M -delete; done
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-pmc > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
python - "$OUT/bench_quick.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "stage_ms")}, "step graph ms", d["decode_step"]["graph_replay_ms"])
PY
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; echo "config5 rc=$?"
python -c "import json; d=json.loads(open('$OUT/bench_config5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'])"
du -sh "$OUT"
