"""Print the top rows of a rocprofv3 kernel_stats csv found under a directory.  python tools/kstats.py <dir> [n]"""
import csv
import glob
import sys

n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('total kernel ms %.3f, launches %d' % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
    for r in rows[:n]:
        print('%9.3f ms %6s calls %9.1f us  %s' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:100]))
