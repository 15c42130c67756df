#!/usr/bin/env python
"""Per-kernel SQ counter summary of a rocprofv3 --pmc pass (GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES) of bench.py.
usage: pmc_sq_summary.py <dir> [title]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); cnt[k] += 1; dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(f"# {sys.argv[2] if len(sys.argv) > 2 else f}\n")
print("SQ_* counters are quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs. "
      "clock = GRBM_GUI_ACTIVE / 8 / duration; VALU busy = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); "
      "the three wave fractions are of SQ_WAVE_CYCLES (parked at s_waitcnt / barrier; issue stall; issuing).\n")
print("| kernel | launches | avg µs | clock GHz | waves | VALU instr / wave | VALU busy | wave: parked | issue stall | issuing |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    n = cnt[k]; wc = v['SQ_WAVE_CYCLES'] or 1; ga = v['GRBM_GUI_ACTIVE'] or 1
    print(f"| `{k}` | {n} | {dur[k]/n:.1f} | {ga/n/8/(dur[k]/n*1e3):.2f} | {v['SQ_WAVES']/n:.0f} | {v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1):.0f} | "
          f"{v['SQ_ACTIVE_INST_VALU']*4/(ga/8*1024):.2f} | {v['SQ_WAIT_ANY']/wc:.2f} | {v['SQ_WAIT_INST_ANY']/wc:.2f} | {v['SQ_ACTIVE_INST_ANY']/wc:.2f} |")
