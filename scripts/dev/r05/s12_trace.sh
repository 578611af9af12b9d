export TMPDIR=/tmp
mkdir -p gpurun_out/r05_dbg
python scripts/dev/ab_variant.py egt "-DPGO_EG_TRACE" 0 -- scripts/dev/timed_region.py C3 3 > /dev/null 2>&1
for v in NONE PGO_EGT_NO_SNAPSHOT PGO_EGT_NO_TIGHT PGO_EGT_NO_SHORT; do
  rm -rf /tmp/tr; env $v=1 PGO_LIBPGO_OVERRIDE=build/variants/libpgo_egt.so rocprofv3 --kernel-trace -d /tmp/tr -o t -- python scripts/dev/timed_region.py C3 20 > gpurun_out/r05_dbg/trace_$v.log 2>&1; echo "$v rc=$? $(grep -c 'graph launch' gpurun_out/r05_dbg/trace_$v.log) graph launches, $(grep -c 'eager' gpurun_out/r05_dbg/trace_$v.log) eager"
done
