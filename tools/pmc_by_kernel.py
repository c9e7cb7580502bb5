"""Reduces rocprofv3 --pmc passes (counter_collection.csv files, --output-format csv) to one line per (kernel, grid): average of
every counter per dispatch, average duration, and the derived clock / matrix-pipe occupancy where the counters allow it.
    python3 tools/pmc_by_kernel.py <dir with *counter_collection.csv> [more dirs ...]"""
import csv, glob, os, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(path) as f:
            for r in csv.DictReader(f):
                name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("pfhip::(anonymous namespace)::", "").replace("void ", ""))
                key = (name[:44], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                did = (path, r["Dispatch_Id"])
                if did not in seen and "Start_Timestamp" in r and r["Start_Timestamp"]:
                    seen.add(did)
                    dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for v in acc.values() for c in v})
print(f"{'kernel':44s} {'blocks':>6s} {'calls':>5s} {'avg_us':>8s} " + " ".join(f"{n[-22:]:>22s}" for n in names) + "   clock_GHz mfma_busy")
for key, v in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]]) if dur[kv[0]] else 0):
    us = sum(dur[key]) / max(1, len(dur[key]))
    line = f"{key[0]:44s} {key[1]:6d} {len(dur[key]):5d} {us:8.1f} " + " ".join(f"{(sum(v[n]) / len(v[n]) if n in v else float('nan')):22.4g}" for n in names)
    if "GRBM_GUI_ACTIVE" in v and us > 0:
        cyc = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]) / 8          # summed over the 8 XCDs
        line += f"   {cyc / (us * 1e3):9.3f}"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            line += f" {sum(v['SQ_VALU_MFMA_BUSY_CYCLES']) / len(v['SQ_VALU_MFMA_BUSY_CYCLES']) / (cyc * 1024):9.3f}"
    print(line)
