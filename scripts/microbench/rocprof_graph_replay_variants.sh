# Characterises scripts/microbench/rocprof_graph_replay_repro (see its header) on a GPU box: plain vs `rocprofv3 --kernel-trace`, by graph size, number of replays and variant bits.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/scripts/microbench/rocprof_graph_replay_repro
O=$GRAFT_REPO_ROOT/gpurun_out/r6_repro_variants.txt
: > $O
run() {   # run <systems> <variant> <every>
  $R $1 $2 $3 > /tmp/p.log 2>&1; a=$?
  rm -rf /tmp/tr; rocprofv3 --kernel-trace -d /tmp/tr -o t -- $R $1 $2 $3 > /tmp/q.log 2>&1; b=$?
  echo "systems $1, chunk of $3 iterations = $((5 * $3)) kernel nodes, variant $2: plain rc=$a ($(tail -1 /tmp/p.log | cut -c5-60)) | rocprofv3 --kernel-trace rc=$b ($(grep -c '^ok' /tmp/q.log) 'ok' line before the end)" >> $O
}
for e in 24 12 4; do for v in 0 15; do run 8 $v $e; done; done
for s in 1 2 3 4; do run $s 15 12; done
for s in 1 2 4 16; do run $s 15 2; done
# the threshold: graphs of 10 kernel nodes, capture up front, nothing eager in between, stopped after R replays
for r in 100 400 800 1200 1500 1700; do
  rm -rf /tmp/tr; rocprofv3 --kernel-trace -d /tmp/tr -o t -- $R 64 15 2 $r > /tmp/q.log 2>&1; b=$?; echo "10-node graph replayed $r times = $((10 * r)) traced graph kernels: rocprofv3 --kernel-trace rc=$b ($(grep -c '^ok' /tmp/q.log) 'ok' line before the end)" >> $O
done
for r in 50 150 250 280; do
  rm -rf /tmp/tr; rocprofv3 --kernel-trace -d /tmp/tr -o t -- $R 64 15 12 $r > /tmp/q.log 2>&1; b=$?; echo "60-node graph replayed $r times = $((60 * r)) traced graph kernels: rocprofv3 --kernel-trace rc=$b ($(grep -c '^ok' /tmp/q.log) 'ok' line before the end)" >> $O
done
rm -rf /tmp/tr; rocprofv3 --hip-trace -d /tmp/tr -o t -- $R 8 0 24 > /tmp/q.log 2>&1; echo "8 systems, 120 nodes, variant 0 under rocprofv3 --hip-trace (no kernel trace): rc=$?" >> $O
rocprofv3 --version 2>&1 | grep "version\|rocm_version" >> $O
