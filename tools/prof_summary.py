#!/usr/bin/env python3
"""Summarise a rocprofv3 --stats kernel_stats.csv: python tools/prof_summary.py <csv> [ncalls_of_sample]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{r['Name'][:80]:80s} calls={int(r['Calls']):6d} ms/sample={float(r['TotalDurationNs'])/1e6/n:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['TotalDurationNs'])/tot*100:5.1f}")
print(f"total kernel ms per sample: {tot/1e6/n:.1f}")
