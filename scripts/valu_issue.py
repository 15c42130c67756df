#!/usr/bin/env python
"""VALU accounting of one bench.py group from a rocprofv3 SQ-counter pass (the counters of scripts/pmc_sq_summary.py):
wave-instructions per op-triple and the time the VALUs spent issuing, written with the hash of the source tree the pass
was measured on (bench.py marks it stale when the kernels have changed).
usage: valu_issue.py <pmc_sq dir> <triples per launch group> [out.json]"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_amd.roofline import csrc_tree_hash  # noqa: E402

f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
group = int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not k.startswith('evah::'):
        continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); cnt[k] += 1; dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
groups = max(cnt[k] for k in cnt if 'ks_inner_kernel' in k)  # the key-switch kernel runs once per group
for k in [k for k in cnt if cnt[k] < groups]:  # set-up kernels (key layout, table builds) are not part of a group
    del agg[k], cnt[k], dur[k]
instr = sum(v['SQ_INSTS_VALU'] for v in agg.values()) / groups / group
kern_us = sum(dur.values()) / groups / group
issue_us = sum(v['SQ_ACTIVE_INST_VALU'] * 4 / (v['GRBM_GUI_ACTIVE'] / 8 * 1024) * dur[k] for k, v in agg.items()) / groups / group
clock = sum(v['GRBM_GUI_ACTIVE'] / 8 for v in agg.values()) / (sum(dur.values()) * 1e3)
out = {"source": "rocprofv3 --pmc SQ_* pass of bench.py (scripts/pmc_sq_summary.py): waves x VALU instructions per wave and VALU-busy x duration, "
                 "summed over the launches of one group of op-triples",
       "tree": csrc_tree_hash(), "commit": os.environ.get("EVA_COMMIT", ""), "kernel_names": sorted(agg),
       "valu_wave_instructions_per_triple": round(instr), "kernel_time_us_per_triple": round(kern_us, 2),
       "valu_issuing_us_per_triple": round(issue_us, 2), "sustained_clock_ghz": round(clock, 2), "simds": 1024}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
