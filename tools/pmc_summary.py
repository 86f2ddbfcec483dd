"""Developer aid: summarise a rocprofv3 counter_collection.csv per kernel (mean over dispatches)."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.OrderedDict()
        for r in rows:
            k = r["Kernel_Name"][:40]
            d = agg.setdefault(k, collections.OrderedDict())
            v = d.setdefault(r["Counter_Name"], [0.0, 0, 0.0])
            v[0] += float(r["Counter_Value"]); v[1] += 1
            v[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, d in agg.items():
            print(path, k)
            for c, (s, n, t) in d.items():
                print("   %-34s %16.0f   (n=%d, avg dur %.1f us)" % (c, s / n, n, t / n / 1e3))
