#!/bin/bash
# Same-box comparison of the round-2 tree (git worktree of the r02 final commit under _r02/, its own library and bench.py) with this tree: bench without events.
export TMPDIR=/tmp
R=$PWD
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0"
for rep in 1 2; do
echo "== B=32 r02"; (cd $R/_r02 && $B --steps 10 --warmup 3 2>&1 | tail -1 | cut -c60-100)
echo "== B=32 r03"; (cd $R && $B --steps 10 --warmup 3 2>&1 | tail -1 | cut -c60-100)
done
for b in 64 8 1; do
  st=$((b == 1 ? 200 : (b == 64 ? 8 : 30)))
  echo "== B=$b r02"; (cd $R/_r02 && $B --batch $b --steps $st --warmup 5 2>&1 | tail -1 | cut -c60-100)
  echo "== B=$b r03"; (cd $R && $B --batch $b --steps $st --warmup 5 2>&1 | tail -1 | cut -c60-100)
done
