#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r05_c9_tests.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r05_c9_smoke.log 2>&1
tail -5 gpurun_out/r05_c9_tests.log; cat gpurun_out/r05_c9_smoke.log
