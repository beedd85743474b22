"""print the kernel sequence of one bench step (name, grid, duration) from a rocprofv3 kernel trace"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# one step = from the kernel after an optimizer launch to the next optimizer launch (inclusive)
idx = [i for i, n in enumerate(names) if 'adamw' in n]
a, b = idx[-3] + 1, idx[-2] + 1
tot = 0
prev_end = None
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = 0 if prev_end is None else (int(r['Start_Timestamp']) - prev_end) / 1e3
    prev_end = int(r['End_Timestamp'])
    tot += d
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:70]:70s} grid={int(r['Grid_Size_X'])//max(int(r['Workgroup_Size_X']),1):6d}x{r['Workgroup_Size_X']:>4s} {d:7.1f}us gap={gap:5.1f}")
print("sum", tot, "wall", (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
