"""Compact view of a rocprofv3 --kernel-trace --stats kernel_stats.csv (names truncated, sorted by total time)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms over {len(rows)} distinct kernels")
print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    n = r["Name"]
    n = n if len(n) <= 70 else n[:67] + "..."
    print(f"{n:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
