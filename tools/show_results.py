import json,sys
for f in sys.argv[1:]:
    for line in open(f):
        line=line.strip()
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], (d.get("config") if isinstance(d.get("config"),str) else "headline text")[:40], d.get("decompressed_MB_per_s", d.get("value")), d.get("kernel_ms", d.get("ms_per_step")))
