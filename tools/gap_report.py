"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: which launches
leave the GPU waiting, per (previous kernel -> next kernel) pair.
usage: python tools/gap_report.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]


def short(name):
    m = re.search(r"(?:mpm::)?(\w+)(?:<[^>]*>)?\(", name)
    return m.group(1) if m else name[:30]


busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = max(0, s1 - e0)
    key = (short(n0), short(n1))
    gaps[key][0] += g
    gaps[key][1] += 1
print("kernels %d  span %.3f ms  busy %.3f ms  idle %.3f ms (%.1f%%)" %
      (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
for key, (total, count) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:16]:
    print("%-28s -> %-28s  n=%5d  avg gap %7.2f us  total %8.3f ms" % (key[0], key[1], count, total / count / 1e3, total / 1e6))
