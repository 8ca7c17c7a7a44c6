"""Dev tool: average PMC counter values per kernel from a rocprofv3 counter_collection.csv."""
import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_" in k and "rocprim" not in k and "bvh" not in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            print(k[:44], " ".join(f"{n}={sum(v)/len(v):.4g}" for n, v in sorted(c.items())))
