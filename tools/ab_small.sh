#!/bin/bash
# A/B of library builds on the small-stream workloads (tests/bench_configs.py --which small,tiny): tools/ab_small.sh lib1.so lib2.so ...
cd "$(dirname "$0")/.."
for lib in "$@"; do
  echo "== $lib"
  DEFL_AMD_LIB=$PWD/$lib timeout -k 10 200 python -u tests/bench_configs.py --which ${AB_WHICH:-small,tiny} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('%-72s %8s ms %8.1f GB/s' % (r['config'][:72], r.get('kernel_ms'), r.get('decompressed_MB_per_s', 0) / 1e3))
" || exit 1
done
