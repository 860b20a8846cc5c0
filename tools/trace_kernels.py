"""List every mrp_* kernel dispatch of a rocprofv3 --kernel-trace run in launch order:
usage: trace_kernels.py <rocprof_out_dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "mrp_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:10.3f} ms  +{(e - s) / 1e6:9.3f} ms  grid {int(r['Grid_Size_X']):>9d} wg {r['Workgroup_Size_X']:>4s} lds {r.get('LDS_Block_Size', '?'):>7s} scratch {r.get('Scratch_Size', '?'):>5s} vgpr {r.get('VGPR_Count', '?'):>4s}  {r['Kernel_Name'].split('(')[0][:48]}")
