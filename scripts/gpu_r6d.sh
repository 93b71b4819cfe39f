#!/bin/bash
# Round 6, fourth GPU call: every GPU test again; wide passes from 128 rows (decode steps only, two column tiles for the first MLP projection) against
# the 16-row tiles; literal configs[2] (4 WebSocket clients, small multilingual, language detection, VAD on) with and without the batch worker; a
# stream with realistic decode lengths (16 tokens per chunk) and where its latency goes.
set -u
TAG=${1:-r6d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)" "$OUT/pytest.log" | head -20
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ds = d.get("decode_step", {})
    print("  ", json.dumps({"value": round(d.get("value"), 1), "ms_per_step": round(d.get("ms_per_step"), 2), "step_rows": ds.get("rows"), "step_ms": ds.get("graph_replay_ms")}))
    ks = sorted(ds.get("kernels", []), key=lambda k: -k["total_us"])[:9]
    for k in ks: print("      %-62s n=%5.1f avg %7.2f tot %8.1f" % (k["name"][:62], k["launches"], k["avg_us"], k["total_us"]))
except Exception as e:
    print("   (no JSON line:", e, ")")
PY
}
for w in 128 0; do
  export WLX_WIDE_MIN_ROWS=$w
  echo "=== WLX_WIDE_MIN_ROWS=$w"
  echo "== small.en batch 48"; timeout 300 $B --batch 48 --steps 2 --warmup 1 > "$OUT/bench_b48_w$w.json" 2> "$OUT/bench_b48_w$w.err"; line "$OUT/bench_b48_w$w.json"
  echo "== config 5 max-batch 32"; timeout 500 python bench.py --config 5 --steps 1 --warmup 1 --max-batch 32 --no-pmc > "$OUT/bench_c5_mb32_w$w.json" 2> "$OUT/bench_c5_mb32_w$w.err"; line "$OUT/bench_c5_mb32_w$w.json"
done
unset WLX_WIDE_MIN_ROWS
sline() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st = d.get("stream", {})
    for k in ("unpaced", "paced_256ms", "stage_ms_per_chunk", "config", "error"):
        if k in st: print("   ", k, json.dumps(st[k]))
except Exception as e:
    print("   (no JSON line:", e, ")")
PY
}
S="python bench.py --no-cpu-baseline --no-throughput --no-pmc --steps 3 --warmup 1"
echo "== configs[2] literal: 4 WebSocket clients, small multilingual, per-client decodes"; timeout 400 $S --model small --stream-clients 4 > "$OUT/bench_c2_clients4.json" 2> "$OUT/bench_c2_clients4.err"; sline "$OUT/bench_c2_clients4.json"
echo "== configs[2] literal: the same through the batch worker"; timeout 400 $S --model small --stream-clients 4 --stream-batch > "$OUT/bench_c2_clients4_batch.json" 2> "$OUT/bench_c2_clients4_batch.err"; sline "$OUT/bench_c2_clients4_batch.json"
echo "== configs[1] stream, 16 tokens per chunk (realistic decode length)"; timeout 400 $S --decode-steps 16 > "$OUT/bench_stream_16tok.json" 2> "$OUT/bench_stream_16tok.err"; sline "$OUT/bench_stream_16tok.json"
echo "== configs[1] stream, 64 tokens per chunk"; timeout 400 $S > "$OUT/bench_stream_64tok.json" 2> "$OUT/bench_stream_64tok.err"; sline "$OUT/bench_stream_64tok.json"
echo "total $(( $(date +%s) - t0 )) s"
