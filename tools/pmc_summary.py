"""Merge rocprofv3 --pmc passes (one counter_collection CSV per pass) into profiles/<tag>_pmc_summary.json:
per kernel the launch count and the average counter value per launch (FETCH_SIZE / WRITE_SIZE are in KB).
usage: python tools/pmc_summary.py out.json [--meta '{"config": "c2", "lanes": 96, "scene": "corridor"}'] pass1_counter_collection.csv [pass2 ...]
(--meta: the launch shape the passes were taken on; bench.py attaches the counters only to runs of that shape)"""
import csv, json, sys, collections, re

def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return name.replace("vslam::", "")

out = collections.defaultdict(dict)
args = sys.argv[2:]
meta = None
if args and args[0] == "--meta":
    meta = json.loads(args[1]); args = args[2:]
for path in args:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            out[k]["launches"] = len(v)
            out[k][c + "_avg"] = sum(v) / len(v)
if meta is not None:
    out["_meta"] = meta
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print("kernels:", len(out))
