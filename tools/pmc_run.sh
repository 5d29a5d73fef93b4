#!/bin/bash
# Collects the PMC passes of the headline bench (one rocprofv3 run per pass) and prints the means per kernel.
#   tools/pmc_run.sh <tag> [python script and arguments]      (on the GPU box, from the repository root)
# Without a script: bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary.
set -u
tag=${1:-pmc}
shift
if [ $# -eq 0 ]; then set -- bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary; fi
prog=("$@")
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
pass() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${tag}_$name -o pmc -- python3 "${prog[@]}" > gpurun_out/${tag}_$name.log 2>&1 || return 1
}
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES &&
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT &&
pass c TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE &&
pass d FETCH_SIZE &&
pass e WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py gpurun_out/${tag}_a gpurun_out/${tag}_b gpurun_out/${tag}_c gpurun_out/${tag}_d gpurun_out/${tag}_e > gpurun_out/${tag}_summary.csv
grep -E "CompactFmt, 3|png" gpurun_out/${tag}_summary.csv || true
