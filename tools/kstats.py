"""Print the top rows of a rocprofv3 kernel_stats.csv: python3 tools/kstats.py <dir> [n]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for row in list(csv.DictReader(open(f)))[:n]:
    print(f"{row['Name'][:72]:72s} calls {row['Calls']:>4s} avg_us {float(row['AverageNs']) / 1e3:9.1f} total_ms {float(row['TotalDurationNs']) / 1e6:8.2f}")
