#!/bin/bash
# Build libwlx.so + libwlx_trace.so (+ named A/B variants) and FAIL LOUDLY: a GPU call after a failed build would measure the old library.
set -e
python - "$@" <<'PY'
import sys
from whisperlive_amd import _lib
_lib.build(force=True); _lib.build_trace()
for v in sys.argv[1:]:
    name, *defs = v.split(":")
    if not (name.startswith("libwlx_") and name.endswith(".so")):
        sys.exit(f"build_all.sh: variant '{v}' must be libwlx_<name>.so:DEFINE[:DEFINE...] (a bare name would write a file nobody loads)")
    _lib.build_variant(name, defs)
print("build ok")
PY
