"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, for the LARGEST dispatches
(the timed replays of bench.py), mean of each counter."""
import csv, sys, collections, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"][:40]
    by[(name, int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)))][r["Counter_Name"]].append(float(r["Counter_Value"]))
# keep for each kernel name the grid size with the largest grid
best = {}
for (name, grid), d in by.items():
    if name not in best or grid > best[name][0]:
        best[name] = (grid, d)
for name, (grid, d) in sorted(best.items()):
    print(f"{name}  grid={grid}")
    for c, v in sorted(d.items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
