cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -m gpu -k "variant or 2 or fuzz" 2>&1 | tail -3 || exit 1
DEFL_AMD_LIB=$GRAFT_REPO_ROOT/defl_amd/libdefl_amd_prof.so timeout -k 10 200 python3 tools/par_profile.py || exit 1
timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --streams 32768 --pool 512 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])" || exit 1
