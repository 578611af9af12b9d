export TMPDIR=/tmp
mkdir -p gpurun_out/r05_dbg
for mode in "--kernel-trace --stats" "--kernel-trace"; do
  rm -rf /tmp/tr; rocprofv3 $mode -d /tmp/tr -o t -- python scripts/dev/timed_region.py C3 20 > gpurun_out/r05_dbg/a.log 2>&1; echo "mode [$mode] rc=$? $(grep steps gpurun_out/r05_dbg/a.log | tail -1)"
done
rm -rf /tmp/tr; rocprofv3 --kernel-trace -d /tmp/tr -o t -- python scripts/dev/r05/ab_options.py C3 20 1 "cg_use_graph=0" > gpurun_out/r05_dbg/b.log 2>&1; echo "no graph rc=$? $(grep it/s gpurun_out/r05_dbg/b.log | tail -1 | cut -c1-100)"
rm -rf /tmp/tr; rocprofv3 --kernel-trace -d /tmp/tr -o t -- python scripts/dev/r05/ab_options.py C3 20 1 "cg_end_game=0" > gpurun_out/r05_dbg/c.log 2>&1; echo "no end game rc=$? $(grep it/s gpurun_out/r05_dbg/c.log | tail -1 | cut -c1-100)"
rm -rf /tmp/tr; rocprofv3 --kernel-trace -d /tmp/tr -o t -- python scripts/dev/r05/ab_options.py C3 20 1 "" > gpurun_out/r05_dbg/d.log 2>&1; echo "defaults via ab_options rc=$? $(grep it/s gpurun_out/r05_dbg/d.log | tail -1 | cut -c1-100)"
