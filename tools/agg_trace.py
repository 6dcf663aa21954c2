"""Aggregate a rocprofv3 kernel_trace.csv by (short kernel name, grid): total µs, calls, avg µs."""
import csv, sys, re, collections
path, div = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    name = re.sub(r"^void ", "", name)
    short = name.split("(")[0][:70]
    if "--grid" in sys.argv:
        short += " g%s" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[short]; a[0] += d; a[1] += 1
tot = sum(a[0] for a in agg.values())
print("total %.1f us (/%g = %.1f)" % (tot, div, tot / div))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 45]:
    print("%10.1f %6d %8.1f  %s" % (a[0] / div, a[1], a[0] / a[1], k))
