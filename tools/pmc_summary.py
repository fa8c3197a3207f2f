"""Mean per-launch PMC counter values per kernel from rocprofv3 --pmc output directories (diagnostic).
usage: pmc_summary.py DIR [DIR ...]   (each DIR holds *_counter_collection.csv of one --pmc pass)"""
import csv, glob, os, re, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)                      # (dispatch, counter) -> sum over dimensions
        name = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per[key] += float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, c), v in per.items():
            vals[name[disp]][c].append(v)
counters = sorted({c for k in vals for c in vals[k]})
def short(n):
    n = re.sub(r"void |\(anonymous namespace\)::|at::native::", "", n)
    return re.sub(r"\(.*", "", n)[:60]
print(",".join(["kernel", "launches"] + ["mean_" + c for c in counters]))
rows = []
for k, cs in vals.items():
    n = max(len(v) for v in cs.values())
    rows.append((-(sum(cs.get(counters[0], [0])) if counters else 0), [short(k), str(n)] + ["%.1f" % (sum(cs[c]) / len(cs[c])) if c in cs else "" for c in counters]))
for _, r in sorted(rows, key=lambda x: x[0])[:45]:
    print(",".join('"%s"' % x if "," in x else x for x in r))
