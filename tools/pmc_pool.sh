#!/bin/bash
# Instruction-cache and stall counters of the headline bench at two pool sizes (distinct streams).
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for p in 2048 8192; do
  timeout -s KILL 170 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmcpool_$p -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --pool $p > gpurun_out/pmcpool_$p.log 2>&1 || exit 1
  echo "pool $p"
  python3 tools/pmc_summary.py gpurun_out/pmcpool_$p | grep "CompactFmt, 3\|CompactFmtELi3E" | sed 's/"[^"]*",/k3,/'
done
