#!/bin/bash
# sweep: workgroups of dm_beam_write (LA3DM_BEAM_WGS) and of dm_grid_centroids_big (LA3DM_BIG_WGS), kernel trace at both sizes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/wgs; mkdir -p $O
for w in 128 256 512 1024; do
  LA3DM_BEAM_WGS=$w bash tools/prof/prof_devmap.sh 1000000 3 0.05 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/tl_1M_beam$w.txt
  LA3DM_BEAM_WGS=$w bash tools/prof/prof_devmap.sh 200000 4 0.1 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/tl_200k_beam$w.txt
done
for w in 4096 8192; do
  LA3DM_BIG_WGS=$w bash tools/prof/prof_devmap.sh 1000000 3 0.05 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/tl_1M_big$w.txt
  LA3DM_BIG_WGS=$w bash tools/prof/prof_devmap.sh 200000 4 0.1 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/tl_200k_big$w.txt
done
for f in $O/tl_*beam*; do echo "$f: $(grep dm_beam_write $f | tail -1 | cut -c1-60)"; done
for f in $O/tl_*beam128* $O/tl_*big*; do echo "$f: $(grep dm_grid_centroids_big $f | tail -2 | cut -c19-32 | tr '\n' ' ')"; done
