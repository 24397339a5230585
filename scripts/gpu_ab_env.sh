#!/bin/bash
# Same-box, same-library A/B of an engine switch read from the environment (AB_ENV="NAME=value" = the OLD form) at B = 32 / 8 / 1.
export TMPDIR=/tmp
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0"
for rep in 1 2; do
echo "== B=32 $AB_ENV"; env $AB_ENV $B --steps 10 --warmup 3 2>&1 | tail -1 | cut -c60-100
echo "== B=32 default"; $B --steps 10 --warmup 3 2>&1 | tail -1 | cut -c60-100
done
for b in 8 1; do
  st=$((b == 1 ? 200 : 30))
  echo "== B=$b $AB_ENV"; env $AB_ENV $B --batch $b --steps $st --warmup 5 2>&1 | tail -1 | cut -c60-100
  echo "== B=$b default"; $B --batch $b --steps $st --warmup 5 2>&1 | tail -1 | cut -c60-100
done
