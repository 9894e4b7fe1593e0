#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
export FUZZ_DEGENERATE=1 FUZZ_DEVICE_TRIG=1
run() { echo "== $*"; env "$@" python tools/fuzz_parity.py 40 457738 2>&1 | grep -E "^CASE|cases,|rror" | cut -c1-200; }
run SVSDF_LIB_VARIANT=base
run FUZZ_PIECE_TIME=fast
