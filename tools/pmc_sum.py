"""per-kernel averages of the counters in rocprofv3 counter_collection.csv files: tools/pmc_sum.py <dir> [name filter]"""
import csv, sys, glob, collections
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(sys.argv[1] + '/*/*/*_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
        if flt and flt not in n: continue
        acc[n][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[n][r['Counter_Name']] += 1
    for n in acc:
        print(n)
        for c in acc[n]:
            print(f"   {c:28s} {acc[n][c] / cnt[n][c]:16.0f}")
