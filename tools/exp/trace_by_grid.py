"""Per (kernel, grid size) totals of a rocprofv3 kernel trace: python tools/exp/trace_by_grid.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    import re
    name = re.sub(r"^void |\(anonymous namespace\)::|rocprim::ROCPRIM_\d+_NS::detail::", "", r["Kernel_Name"]).split("(")[0][:44]
    key = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))
    acc[key][0] += 1
    acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]
for (name, grid), (n, us) in rows:
    print(f"{name:40s} grid {grid:9d} calls {n:6d} total {us / 1e3:8.2f} ms avg {us / n:8.1f} us")
