timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for v in 1 2; do
  echo "== variant $v lanes 16"
  timeout -k 10 150 python -u bench.py --streams 32768 --pool 512 --steps 3 --warmup 1 --lanes 16 --variant $v --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
