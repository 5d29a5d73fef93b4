export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/r3; mkdir -p $OUT/r3
( while sleep 45; do echo "[hb] $(date +%T)"; done ) &
HB=$!
trap "kill $HB" EXIT
cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline"
echo "[trace]"; timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3/kt -o p -- python3 $ARGS > $OUT/r3/kt.log 2>&1 || { grep "bench " $OUT/r3/kt.log; exit 1; }
echo "[pmc1]"; timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/r3/pmc1 -o p -- python3 $ARGS > $OUT/r3/pmc1.log 2>&1 || { grep "bench " $OUT/r3/pmc1.log; exit 1; }
echo "[pmc2]"; timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/r3/pmc2 -o p -- python3 $ARGS > $OUT/r3/pmc2.log 2>&1 || { grep "bench " $OUT/r3/pmc2.log; exit 1; }
echo "[pmc3]"; timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/r3/pmc3 -o p -- python3 $ARGS > $OUT/r3/pmc3.log 2>&1 || { grep "bench " $OUT/r3/pmc3.log; exit 1; }
echo "[pmc4]"; timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE SQ_BUSY_CYCLES --output-format csv -d $OUT/r3/pmc4 -o p -- python3 $ARGS > $OUT/r3/pmc4.log 2>&1 || { grep "bench " $OUT/r3/pmc4.log; exit 1; }
python3 - <<'PY'
import csv, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3'
rows=[]
for d in ('pmc1','pmc2','pmc3','pmc4'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f'{out}/{d}/p_counter_collection.csv')):
        if 'infl_batch_kernel' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): rows.append((k[0],k[1],sum(v)/len(v),len(v)))
with open(out+'/pmc_summary.csv','w') as f:
    f.write('kernel,counter,mean_per_launch,launches\n')
    for r in rows: f.write('"%s",%s,%.1f,%d\n'%r)
print(open(out+'/pmc_summary.csv').read())
PY
cat $OUT/r3/kt/p_kernel_stats.csv | cut -c1-200
