export TMPDIR=/tmp
OUT=gpurun_out/r6_shipped_trace.txt
: > $OUT
for steps in 8 10; do
  rm -rf /tmp/trs; rocprofv3 --kernel-trace -d /tmp/trs -o t -- python scripts/dev/timed_region.py C3 $steps > /tmp/ts.log 2>&1; rc=$?
  echo "## rocprofv3 --kernel-trace -- python scripts/dev/timed_region.py C3 $steps      (library defaults: PCG chunks replayed as hipGraphs — the SHIPPED configuration; rc=$rc)" >> $OUT
  if [ $rc -eq 0 ]; then python scripts/rocpd_summary.py segments $(find /tmp/trs -name "*.db" | head -1) 50 200 >> $OUT; tail -1 /tmp/ts.log >> $OUT; else tail -3 /tmp/ts.log | cut -c1-200 >> $OUT; fi
done
