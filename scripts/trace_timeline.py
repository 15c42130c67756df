"""Timeline of the last `n` dispatches of a rocprofv3 kernel-trace CSV: start, duration, gap to the previous kernel's end,
grid, VGPRs, name.   python scripts/trace_timeline.py <kernel_trace.csv> [n] [name filter to start from]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
last = rows[-n:]
t0 = int(last[0]['Start_Timestamp'])
prev_end, tot, gaps = None, 0.0, 0.0
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1000 if prev_end is not None else 0.0
    name = r['Kernel_Name'].replace('evah::', '').replace('void ', '')[:80]
    wg = int(r['Workgroup_Size_X'])
    print(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:8.1f} gap {gap:7.1f} q{r['Queue_Id']} grid {int(r['Grid_Size_X']) // wg}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg{wg} vgpr{r['VGPR_Count']} {name}")
    prev_end = max(e, prev_end or e)
    tot += (e - s) / 1000
    gaps += max(gap, 0.0)
print(f"kernel sum {tot:.1f} us, positive gaps {gaps:.1f} us, span {(prev_end - t0) / 1000:.1f} us")
