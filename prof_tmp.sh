export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/r1
cd $GRAFT_REPO_ROOT
echo "[bench full]"; timeout -k 10 400 python3 -u bench.py --steps 10 --warmup 2 > $OUT/r1/bench_full.json 2> $OUT/r1/bench_full.err || exit 1
cat $OUT/r1/bench_full.json
ARGS="bench.py --steps 5 --warmup 1 --pool 512 --no-cpu-baseline"
echo "[kernel trace]"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r1/kt -o kt -- python3 $ARGS > $OUT/r1/kt.log 2>&1 || exit 1
echo "[pmc1]"; timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/r1/pmc1 -o p -- python3 $ARGS > $OUT/r1/pmc1.log 2>&1 || exit 1
echo "[pmc2]"; timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/r1/pmc2 -o p -- python3 $ARGS > $OUT/r1/pmc2.log 2>&1 || exit 1
echo "[pmc3 fetch]"; timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/r1/pmc3 -o p -- python3 $ARGS > $OUT/r1/pmc3.log 2>&1 || exit 1
echo "[pmc4 write]"; timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/r1/pmc4 -o p -- python3 $ARGS > $OUT/r1/pmc4.log 2>&1 || exit 1
find $OUT/r1 -name "*.csv" | head -30
