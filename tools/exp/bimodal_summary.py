"""Per (kernel variant, grid) means of duration and counters from one rocprofv3 --pmc pass: python bimodal_summary.py <dir> <tag>"""
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    m = re.search(r"k_sell_spmv<([^>]*)>", r["Kernel_Name"])
    if not m:
        continue
    key = (m.group(1).replace(" ", ""), int(r["Grid_Size"]))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for key in sorted(acc, key=lambda k: -k[1])[:3]:
    d = list(dur[key].values())
    line = f"process {sys.argv[2]} <{key[0]}> grid {key[1]} n {len(d)} avg_us {sum(d) / len(d):8.1f}"
    for c in sorted(acc[key]):
        v = acc[key][c]
        line += f" | {c} {sum(v) / len(v):.4g}"
    print(line)
