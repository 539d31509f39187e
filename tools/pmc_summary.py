"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel average counter value per launch."""
import csv, sys, json, collections, re
out = {}
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = dict(launches=len(v), avg=sum(v) / len(v))
print(json.dumps(out, indent=1))
