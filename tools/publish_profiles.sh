#!/bin/bash
# Copies what tools/collect_profiles.sh <tag> left under gpurun_out/ into profiles/ and rewrites profiles/traffic.json
# from the PMC passes:  tools/publish_profiles.sh <tag>
set -e
tag=${1:-rXX}
cd "$(dirname "$0")/.."
g=gpurun_out/$tag
cp ${g}_bench.json profiles/${tag}_bench.json
cp ${g}_trace/trace_kernel_stats.csv profiles/${tag}_kernel_stats_all_workloads.csv
cp ${g}_trace_headline/trace_kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp ${g}_pmc_summary.csv profiles/${tag}_pmc_summary.csv
cp ${g}_phase_cycles.txt profiles/${tag}_phase_cycles.txt
cp ${g}_png_bench.json profiles/${tag}_png_bench.json
cp ${g}_pngpmc_summary.csv profiles/${tag}_png_pmc_summary.csv
cp ${g}_ubench.txt profiles/${tag}_ubench.txt
[ -s ${g}_configs.jsonl ] && cp ${g}_configs.jsonl profiles/${tag}_configs.jsonl
python3 - "$tag" <<'PY'
import csv, json, sys
tag = sys.argv[1]
rows = {}
for r in csv.reader(open("profiles/%s_pmc_summary.csv" % tag)):
    if "CompactFmt, 3" in r[0]:
        rows[r[1]] = float(r[2])
f, w = rows["FETCH_SIZE"], rows["WRITE_SIZE"]
d = json.load(open("profiles/traffic.json"))
d["hbm_bytes_per_launch"] = int((f + w) * 1024)
d["note"] = ("FETCH_SIZE %.0f KiB + WRITE_SIZE %.0f KiB per launch of infl_batch_kernel<64,CompactFmt,3> (rocprofv3 --pmc, separate "
             "passes, profiles/%s_pmc_summary.csv; the counters are KiB). WRITE_SIZE is taken as exact for this store pattern (4.29 GB of "
             "output plus result arrays); FETCH_SIZE is taken as printed: the x2 correction of gfx950 applies to 16-byte-per-lane streaming "
             "reads, these reads are 4-byte staging loads and byte gathers (uncalibrated widths), so read traffic lies between the printed "
             "figure and twice that." % (f, w, tag))
json.dump(d, open("profiles/traffic.json", "w"), indent=1)
out = 65536 * 65536
print("traffic", d["hbm_bytes_per_launch"], "VALU/byte %.3f SALU/byte %.3f" % (rows["SQ_INSTS_VALU"] / out, rows["SQ_INSTS_SALU"] / out))
PY
