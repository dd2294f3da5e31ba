"""Summarise a rocprofv3 --kernel-trace --stats CSV directory (developer tool)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches ({f})")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}% calls={r['Calls']:>6} avg={float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:100]}")
