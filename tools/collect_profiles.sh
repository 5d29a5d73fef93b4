#!/bin/bash
# Collects what profiles/ holds for a round, on the GPU box, from the repository root:
#   tools/collect_profiles.sh <tag>
# the bench line, the rocprofv3 kernel trace of the same command, the PMC passes of the bench and of the
# PNG un-filter timing, the micro-benchmarks.  Every step has its own time limit; the first failure ends the run.
set -u
tag=${1:-rXX}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
o=gpurun_out/$tag
echo "[collect] bench" &&
timeout -k 10 400 python3 -u bench.py > ${o}_bench.json 2> ${o}_bench.err &&
echo "[collect] kernel trace" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${o}_trace -o trace -- python3 bench.py > ${o}_trace_bench.json 2> ${o}_trace.err &&
echo "[collect] kernel trace, headline workload alone" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${o}_trace_headline -o trace -- python3 bench.py --no-secondary --no-cpu-baseline > ${o}_trace_headline_bench.json 2> ${o}_trace_headline.err &&
echo "[collect] pmc bench" &&
bash tools/pmc_run.sh ${tag}_pmc > ${o}_pmc.out 2>&1 &&
echo "[collect] png" &&
timeout -k 10 200 python3 -u tools/png_bench.py > ${o}_png_bench.json 2> ${o}_png_bench.err &&
bash tools/pmc_run.sh ${tag}_pngpmc tools/png_bench.py --reps 2 > ${o}_pngpmc.out 2>&1 &&
echo "[collect] phase cycles" &&
DEFL_AMD_LIB=$PWD/defl_amd/libdefl_amd_prof.so timeout -k 10 200 python3 -u tools/par_profile.py > ${o}_phase_cycles.txt 2> ${o}_phase_cycles.err &&
echo "[collect] ubench" &&
timeout -k 10 200 tools/ubench/ubench > ${o}_ubench.txt 2>&1 &&
echo "[collect] done"
