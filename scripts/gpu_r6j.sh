#!/bin/bash
# Round 6: the stream leg after the host-side trim (non_speech_tokens memoised): 16 and 64 tokens per chunk, stage shares.
set -u
TAG=${1:-r6j}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
sline() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d.get("stream", {})
print("   headline", round(d["value"], 1), round(d["ms_per_step"], 2))
for k in ("unpaced", "paced_256ms", "stage_ms_per_chunk"):
    if k in st: print("   ", k, json.dumps(st[k]))
PY
}
S="python bench.py --no-cpu-baseline --no-throughput --no-pmc --steps 5 --warmup 2"
echo "== 16 tokens per chunk"; timeout 400 $S --decode-steps 16 > "$OUT/bench_stream_16tok.json" 2> "$OUT/bench_stream_16tok.err"; sline "$OUT/bench_stream_16tok.json"
echo "== 64 tokens per chunk"; timeout 400 $S > "$OUT/bench_stream_64tok.json" 2> "$OUT/bench_stream_64tok.err"; sline "$OUT/bench_stream_64tok.json"
