"""Groups a rocprofv3 kernel_trace.csv by (kernel, grid) and prints calls / avg / total duration."""
import csv
import re
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("pfhip::(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
               r["VGPR_Count"], r["LDS_Block_Size"])
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':58s} {'blocks':>7s} {'gy':>3s} {'gz':>3s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s}")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0][:58]:58s} {k[1]:7d} {k[2]:3d} {k[3]:3d} {k[4]:>5s} {k[5]:>6s} {len(v):6d} {sum(v)/len(v)/1e3:9.1f} {sum(v)/1e6:9.2f} {100*sum(v)/tot:6.2f}")
