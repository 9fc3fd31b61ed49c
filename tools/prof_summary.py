"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel total ms per bench step."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total GPU ms per step: %.2f" % (tot / 1e6 / steps))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-100s calls/step %7.1f ms/step %8.3f avg_us %8.1f %5.1f%%" % (r['Name'][:100], float(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps,
                                                                        float(r['AverageNs']) / 1e3, float(r['Percentage'])))
