#!/bin/bash
# Round 6: the one-pass prompt prefill with (WLX_PREFILL_SLABS=1, a switch that existed for this A/B only) and without the K-split MLP projection.
export TMPDIR=/tmp WLX_QUIET=1
for v in 0 1 0 1; do echo "== WLX_PREFILL_SLABS=$v"; WLX_PREFILL_SLABS=$v timeout 200 python scripts/prefill_time.py small.en 2>&1 | grep "small.en"; done
for v in 0 1; do echo "== WLX_PREFILL_SLABS=$v large-v3"; WLX_PREFILL_SLABS=$v timeout 300 python scripts/prefill_time.py large-v3 2>&1 | grep "large-v3"; done
