"""Groups a rocprofv3 kernel_trace.csv by (kernel, grid) and prints calls / avg / total duration."""
import csv
import re
import sys
from collections import defaultdict

def records(path):
    """csv (kernel_trace.csv) or the rocpd sqlite database rocprofv3 writes by default (results.db)."""
    if path.endswith(".db"):
        import sqlite3
        con = sqlite3.connect(path)
        for r in con.execute("select name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, vgpr_count, lds_size, start, end from kernels"):
            yield {"Kernel_Name": r[0], "Grid_Size_X": r[1], "Grid_Size_Y": r[2] // max(1, r[5]), "Grid_Size_Z": r[3] // max(1, r[6]),
                   "Workgroup_Size_X": r[4], "VGPR_Count": str(r[7]), "LDS_Block_Size": str(r[8]), "Start_Timestamp": r[9], "End_Timestamp": r[10]}
    else:
        with open(path) as f:
            yield from csv.DictReader(f)


rows = defaultdict(list)
if True:
    for r in records(sys.argv[1]):
        name = r["Kernel_Name"].replace("pfhip::(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
               r["VGPR_Count"], r["LDS_Block_Size"])
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':58s} {'blocks':>7s} {'gy':>3s} {'gz':>3s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s}")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0][:58]:58s} {k[1]:7d} {k[2]:3d} {k[3]:3d} {k[4]:>5s} {k[5]:>6s} {len(v):6d} {sum(v)/len(v)/1e3:9.1f} {sum(v)/1e6:9.2f} {100*sum(v)/tot:6.2f}")
