#!/bin/bash
# Final tree of round 6, the configurations beside the headline: the stream leg at 16 and 64 tokens per chunk, literal configs[2]
# (4 WebSocket clients, Whisper-small multilingual) with and without the batch worker, four concurrent streams, 12 / 48 windows per decode,
# config 5 (large-v3, 64 clips through the batch worker) at 16 clips per decode, tiny.en and large-v3 single stream.
set -u
TAG=${1:-r6final_others}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ds = d.get("decode_step", {})
    print("  ", json.dumps({"value": round(d.get("value"), 1), "ms_per_step": round(d.get("ms_per_step"), 2), "p50": round(d.get("p50_chunk_latency_ms") or 0, 2), "conditioned": d.get("value_conditioned"), "stage_ms": d.get("stage_ms"), "step_rows": ds.get("rows"), "step_ms": ds.get("graph_replay_ms")}))
except Exception as e:
    print("   (no JSON line:", e, ")")
PY
}
sline() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st = d.get("stream", {})
    for k in ("unpaced", "paced_256ms", "stage_ms_per_chunk", "error"):
        if k in st: print("   ", k, json.dumps(st[k]))
except Exception as e:
    print("   (no JSON line:", e, ")")
PY
}
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
S="python bench.py --no-cpu-baseline --no-throughput --no-pmc --steps 3 --warmup 1"
echo "== configs[1] stream, 16 tokens per chunk"; timeout 400 $S --decode-steps 16 > "$OUT/bench_stream_16_tokens_per_chunk.json" 2> "$OUT/e1.err"; sline "$OUT/bench_stream_16_tokens_per_chunk.json"
echo "== configs[1] stream, 64 tokens per chunk"; timeout 400 $S > "$OUT/bench_stream_64_tokens_per_chunk.json" 2> "$OUT/e2.err"; sline "$OUT/bench_stream_64_tokens_per_chunk.json"
echo "== configs[2] literal, per-client decodes"; timeout 400 $S --model small --stream-clients 4 > "$OUT/bench_configs2_4clients.json" 2> "$OUT/e3.err"; sline "$OUT/bench_configs2_4clients.json"
echo "== configs[2] literal, batch worker"; timeout 400 $S --model small --stream-clients 4 --stream-batch > "$OUT/bench_configs2_4clients_batch_worker.json" 2> "$OUT/e4.err"; sline "$OUT/bench_configs2_4clients_batch_worker.json"
echo "== small.en, 4 streams"; timeout 300 $B --streams 4 --steps 10 --warmup 3 > "$OUT/bench_s4.json" 2> "$OUT/e5.err"; line "$OUT/bench_s4.json"
for b in 12 48; do echo "== small.en batch $b"; timeout 300 $B --batch $b --steps 3 --warmup 1 > "$OUT/bench_b$b.json" 2> "$OUT/e6.err"; line "$OUT/bench_b$b.json"; done
echo "== tiny.en"; timeout 300 $B --model tiny.en --steps 10 --warmup 3 > "$OUT/bench_tiny_en.json" 2> "$OUT/e7.err"; line "$OUT/bench_tiny_en.json"
echo "== large-v3"; timeout 400 $B --model large-v3 --steps 5 --warmup 2 > "$OUT/bench_large_v3.json" 2> "$OUT/e8.err"; line "$OUT/bench_large_v3.json"
echo "== config 5 max-batch 16"; timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --max-batch 16 --no-pmc > "$OUT/bench_config5_mb16.json" 2> "$OUT/e9.err"; line "$OUT/bench_config5_mb16.json"
rm -f "$OUT"/e?.err
echo "total $(( $(date +%s) - t0 )) s"
echo "== in-kernel timeline of one decode step (libwlx_trace.so)"; WLX_LIB=whisperlive_amd/libwlx_trace.so timeout 300 python scripts/trace_step.py --model small.en --t 33 > "$OUT/decode_step_trace.txt" 2>&1; sed -n 1,9p "$OUT/decode_step_trace.txt" | cut -c1-170
echo "total $(( $(date +%s) - t0 )) s"
