cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -u -m pytest tests -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 400 python3 bench.py 2>gpurun_out/bench_err.log | tee gpurun_out/bench_r01_full.json || exit 1
timeout -k 10 500 python3 tools/bench_configs.py --which stored,fixed,mixed,runs,noisy,png 2>/dev/null | tee gpurun_out/configs_r01.jsonl | cut -c1-250
