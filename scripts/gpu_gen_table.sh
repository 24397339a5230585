set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1700 python scripts/gen_tile_table.py --out gpurun_out/gfx950_tiles.txt --batches 1,8,32,64 2>&1 | tail -20
BENCH="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
echo "== shipped table"; for i in 1 2; do $BENCH 2>&1 | tail -1 | cut -c1-120; done
echo "== new table"; export PF_TILE_TABLE=$PWD/gpurun_out/gfx950_tiles.txt; for i in 1 2; do $BENCH 2>&1 | tail -1 | cut -c1-120; done
for b in 1 8; do timeout 100 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-120; done
unset PF_TILE_TABLE
for b in 1 8; do timeout 100 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-120; done
