"""Per-kernel averages of arbitrary PMC counters from a rocprofv3 counter_collection.csv tree."""
import csv, glob, os, sys, collections
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-44:] + " g" + r["Grid_Size"]
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[name] += 1
for name in sorted(tot, key=lambda n: -sum(tot[n].values()))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(name, "n", cnt[name], {k: round(v / cnt[name], 1) for k, v in sorted(tot[name].items())})
