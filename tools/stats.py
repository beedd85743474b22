import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time/step = {tot/steps/1e3:.1f} us")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print(f"{r['Name'][:84]:84s} n/step={int(r['Calls'])/steps:5.1f} avg={float(r['AverageNs'])/1e3:8.1f}us per-step={float(r['TotalDurationNs'])/steps/1e3:8.1f}us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
