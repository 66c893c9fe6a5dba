"""summarise rocprofv3 --pmc CSVs: python tools/pmc_summary.py <dir> [kernel substring]"""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "render_kernel"
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = r["Kernel_Name"][:60]
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(f)
    for k, v in dur.items():
        print("  %s: %.3f ms avg" % (k, sum(v) / len(v)))
    for (k, c), v in sorted(agg.items()):
        print("    %-40s %-26s %.6g" % (k[-40:], c, sum(v) / len(v)))
