#!/usr/bin/env python3
"""Times infl_stream() fed one byte per call over the reference's png_simulation fixture (what bench.py reports in
its secondary list), alone, so that a kernel trace of the run shows the incremental kernel's own duration."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from defl_amd import capi  # noqa: E402

eng = capi.load()
assert eng.lib.infl_gpu_init(0) == 0
bench.secondary_stream_bytewise(eng)            # warm-up
print(json.dumps(bench.secondary_stream_bytewise(eng)))
