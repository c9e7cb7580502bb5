"""Turns two rocprofv3 PMC passes over the same bench command (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE;
MI355X_MICROARCH.md: the two do not fit one pass) into profiles/<round>/pmc_hbm_traffic.json.

    python3 tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B for 16-B/lane reads, same guide); both counters are in KB.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    if path.endswith(".db"):      # rocpd sqlite database (rocprofv3 default output)
        import sqlite3
        con = sqlite3.connect(path)
        rows = [{"Counter_Name": r[0], "Kernel_Name": r[1], "Grid_Size": r[2], "Workgroup_Size": r[3], "Counter_Value": r[4]} for r in
                con.execute("select counter_name, kernel_name, grid_size, workgroup_size, value from counters_collection")]
    else:
        with open(path) as f:
            rows = list(csv.DictReader(f))
    if True:
        for r in rows:
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("pfhip::(anonymous namespace)::", "").replace("void ", "")
            name = re.sub(r"\(.*", "", name)
            blocks = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            acc[(name, blocks)].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels, gemm_bytes, gemm_launches = [], 0.0, 0
    for key in sorted(fetch, key=lambda k: -sum(fetch[k])):
        f = fetch[key]
        w = write.get(key, [0.0])
        f_mb = 2.0 * sum(f) / len(f) / 1024.0
        w_mb = sum(w) / len(w) / 1024.0
        kernels.append({"kernel": key[0], "blocks": key[1], "calls": len(f), "fetch_size_kb_raw": sum(f) / len(f),
                        "fetch_mb_x2": f_mb, "write_mb": w_mb})
        if key[0].startswith(("gemm_f32_", "gemm_p3_")):
            gemm_bytes += (f_mb + w_mb) * 1024 * 1024 * len(f)
            gemm_launches += len(f)
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 "
                   "--warmup 1`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); "
                   "counts L2->fabric requests, Infinity-Cache hits included",
           "gemm_launches": gemm_launches, "gemm_avg_bytes_per_launch": gemm_bytes / max(1, gemm_launches), "kernels": kernels}
    with open(sys.argv[3], "w") as fo:
        json.dump(out, fo, indent=1)
    print(f"gemm: {gemm_launches} launches, {out['gemm_avg_bytes_per_launch'] / 1e6:.1f} MB per launch")


if __name__ == "__main__":
    main()
