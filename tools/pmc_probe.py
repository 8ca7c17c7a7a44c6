"""Mean of every collected counter per frame kernel (1920x1080 launches) from one rocprofv3 --pmc result directory:
  rocprofv3 --pmc C1 C2 ... -d DIR -o pmc -- python3 bench.py --no-cpu-baseline --steps 12 --warmup 2
  python tools/pmc_probe.py DIR"""
import collections, glob, json, os, sqlite3, sys

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
if not f:
    sys.exit("no result database under " + d)
con = sqlite3.connect(f[0])
rows = list(con.execute("select kernel_name, counter_name, value, grid_size from counters_collection"))
short = lambda n: n.split("(")[0].replace("void ", "").strip()
gmin = {}
for k, c, v, g in rows:
    gmin[short(k)] = min(gmin.get(short(k), g), g)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v, g in rows:
    if short(k).startswith("k_") and g == gmin[short(k)]:
        agg[short(k)][c].append(v)
out = {k: {c: round(sum(v) / len(v), 3) for c, v in cs.items()} for k, cs in agg.items() if k.split("<")[0] in ("k_raycast", "k_generate_candidate", "k_spatial_gather", "k_spatial_coop", "k_resolve", "k_gather", "k_raycast_quad")}
print(json.dumps(out, indent=1))
