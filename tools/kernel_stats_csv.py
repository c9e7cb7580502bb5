"""rocprofv3's rocpd database (results.db of a --kernel-trace run) -> the per-kernel `--stats` summary as CSV
(name, calls, total ns, average ns, percentage), the form committed under profiles/."""
import csv
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
acc = defaultdict(list)
for name, start, end in con.execute("select name, start, end from kernels"):
    acc[name].append(end - start)
tot = sum(sum(v) for v in acc.values())
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 4), min(v), max(v)])
