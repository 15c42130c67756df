"""Summarise a rocprofv3 --kernel-trace run (csv or rocpd .db) into a markdown table.
usage: python scripts/rocprof_summary.py <dir> [title] > profiles/rNN_xxx.md"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    c = sqlite3.connect(path)
    suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    q = f"select s.kernel_name, d.end-d.start from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id"
    return [(n, float(t)) for n, t in c.execute(q)]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    return out


def main():
    d = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else d
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        rows += from_db(p)
    if not rows:
        for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            rows += from_csv(p)
    agg = defaultdict(list)
    for n, t in rows:
        agg[n].append(t)
    tot = sum(sum(v) for v in agg.values())
    print(f"# {title}\n")
    print(f"rocprofv3 --kernel-trace; {len(rows)} dispatches, {tot/1e6:.3f} ms total kernel time\n")
    print("| kernel | calls | avg us | min us | max us | total ms | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = n.replace("_ZN4evah", "").replace(".kd", "")
        print(f"| `{short[:100]}` | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {sum(v)/1e6:.3f} | {100*sum(v)/tot:.1f} |")


if __name__ == "__main__":
    main()
