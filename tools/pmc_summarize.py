"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_summarize.py <kernel substring> <dir>..."""
import csv, glob, os, sys, collections
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list)
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"]:
                continue
            per[(r["Dispatch_Id"], r["Counter_Name"])] = float(r["Counter_Value"])
        for (disp, name), v in per.items():
            acc[name].append(v)
        for name in sorted(acc):
            vals = acc[name][3:] if len(acc[name]) > 6 else acc[name]     # skip warm-up launches
            print("%-16s %-28s launches %3d mean %.6g" % (os.path.basename(d), name, len(vals), sum(vals) / len(vals)))
