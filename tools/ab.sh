#!/bin/bash
# A/B timing of library builds on the headline batch in one session: tools/ab.sh lib1.so lib2.so ...
# (each library twice, interleaved; DEFL_AMD_LIB selects the library bench.py loads)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2; do
  for lib in "$@"; do
    printf "%s round %d: " "$lib" $round
    DEFL_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --pool ${AB_POOL:-1024} --steps ${AB_STEPS:-10} --warmup 2 --no-cpu-baseline 2>gpurun_out/ab_err.txt \
      | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'], r['roofline']['kernel_ms'])" || { tail -5 gpurun_out/ab_err.txt; exit 1; }
  done
done
