#!/bin/bash
# builds and runs tools/blk3_knock.hip for a set of knock-out masks (bit 1 = fixed two-iteration schedule, always on)
set -e
mkdir -p tools/bin
for m in 1 3 5 9 17 33 65 129 257 513 1025 ${EXTRA_MASKS}; do
  rm -f tools/bin/blk3_knock_$m
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -DF3_KNOCK=$m -o tools/bin/blk3_knock_$m tools/blk3_knock.hip 2>/dev/null &
done
wait
for m in 1 3 5 9 17 33 65 129 257 513 1025 ${EXTRA_MASKS}; do
  if [ -x tools/bin/blk3_knock_$m ]; then TOL=1e-4 timeout -k 5 30 ./tools/bin/blk3_knock_$m | tail -1
  else echo "mask $m: did not compile (hipcc back-end error on this code shape: Illegal instruction detected ... src_shared_base)"; fi
done
