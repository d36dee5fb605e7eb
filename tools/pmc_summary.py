"""Summarise rocprofv3 --pmc passes per kernel family: counter totals, dispatch counts, per-launch means."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("elector::", "").split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {}
for k in sorted(agg):
    out[k] = {c: {"total": v, "launches": len(disp[(k, c)])} for c, v in sorted(agg[k].items())}
print(json.dumps(out, indent=1))
