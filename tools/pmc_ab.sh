#!/bin/bash
# Instruction and wait counters (two rocprofv3 passes) of the headline batch for each library given:
#   tools/pmc_ab.sh tag1=lib1.so tag2=lib2.so ...     (on the GPU box, from the repository root)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for spec in "$@"; do
  tag=${spec%%=*}; lib=${spec#*=}
  export DEFL_AMD_LIB=$PWD/$lib
  timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmcab_${tag}_a -o pmc -- python3 bench.py --pool 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcab_${tag}_a.log 2>&1 || exit 1
  timeout -s KILL 170 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmcab_${tag}_b -o pmc -- python3 bench.py --pool 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcab_${tag}_b.log 2>&1 || exit 1
  python3 tools/pmc_summary.py gpurun_out/pmcab_${tag}_a gpurun_out/pmcab_${tag}_b | grep -E "CompactFmt, 3" | sed "s/^/$tag /" | tee -a gpurun_out/pmcab_summary.txt
done
