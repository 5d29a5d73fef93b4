#!/usr/bin/env python3
"""One line per JSON result line of bench.py / tests/bench_configs.py output files: file, workload,
decompressed MB/s, milliseconds.  `python tools/show_results.py file.jsonl [more ...]`"""
import json
import sys

for f in sys.argv[1:]:
    for line in open(f):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        name = d.get("config") if isinstance(d.get("config"), str) else "headline text"
        rate = d.get("decompressed_MB_per_s", d.get("value"))
        if rate is None and "gather_GB_per_s_read_plus_written" in d:
            rate = "%s GB/s read + written" % d["gather_GB_per_s_read_plus_written"]
        print(f.split("/")[-1], name[:60], rate,
              d.get("kernel_ms", d.get("ms_per_step", d.get("call_ms", d.get("infl_many_ms", d.get("gather_ms"))))))
