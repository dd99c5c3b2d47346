#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
sed -i 's/dump_traceback_later(90/dump_traceback_later(40/' tools/dbg_small_verify.py
for i in $(seq 1 40); do
  timeout 80 python tools/dbg_small_verify.py > gpurun_out/dbg_loop.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "iteration $i rc=$rc"; tail -60 gpurun_out/dbg_loop.log; cp gpurun_out/dbg_loop.log gpurun_out/dbg_loop_fail.log; break; fi
done
echo "done $i"
