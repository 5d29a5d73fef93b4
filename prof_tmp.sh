export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/r2
( while sleep 45; do echo "[hb] $(date +%T)"; done ) &
HB=$!
trap "kill $HB" EXIT
cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 3 --warmup 1 --streams 32768 --pool 512 --variant 2 --no-cpu-baseline"
echo "[pmc1]"; timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/r2/pmc1 -o p -- python3 $ARGS > $OUT/r2/pmc1.log 2>&1 || { grep "bench " $OUT/r2/pmc1.log; exit 1; }
echo "[pmc2]"; timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/r2/pmc2 -o p -- python3 $ARGS > $OUT/r2/pmc2.log 2>&1 || { grep "bench " $OUT/r2/pmc2.log; exit 1; }
echo "[pmc3]"; timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/r2/pmc3 -o p -- python3 $ARGS > $OUT/r2/pmc3.log 2>&1 || { grep "bench " $OUT/r2/pmc3.log; exit 1; }
python3 - <<'PY'
import csv, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2'
for d in ('pmc1','pmc2','pmc3'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f'{out}/{d}/p_counter_collection.csv')):
        if 'infl_batch_kernel' in r['Kernel_Name'] and 'Li3E' in r['Kernel_Name'] or 'CompactFmt, 3' in r['Kernel_Name']:
            agg[(r['Counter_Name'], r['VGPR_Count'], r['LDS_Block_Size'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(d,k,sum(v)/len(v),len(v))
PY
