"""Print a compact per-kernel table from a rocprofv3 kernel_stats csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:8.2f} pct={r['Percentage']}")
