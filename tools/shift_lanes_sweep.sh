#!/bin/bash
# Where the one-pair-per-wavefront shift kernel should hand over to the one-pair-per-lane rotation kernel: M2M / L2L stage times of
# the bench workload for one GPU and for one rank of eight, per order, for several values of FMMBEM_SHIFT_LANES_MAX (pair-slots per
# launch up to which the wavefront kernel runs; 0 = never).  usage: tools/shift_lanes_sweep.sh > out.txt
cd "$(dirname "$0")/.."
for p in 2 4 6 8 10 12; do
  for t in 0 2048 8192 16384 65536 1000000; do
    echo -n "p=$p max=$t  "
    FMMBEM_SHIFT_LANES_MAX=$t python tools/shard_time.py --p $p --worlds 1,8 --steps 10 2>/dev/null | grep world | sed -e 's/back to back.*| near [0-9.]* p2m [0-9.]*//' -e 's/m2l.*l2l/l2l/' -e 's/l2p.*//' | tr '\n' ' '
    echo
  done
done
