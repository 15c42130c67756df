"""Total HBM traffic of a DAG leg from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of scripts/prof_legs.py:
sums the counters over every kernel of the run and divides by the number of executions the leg made.
gfx950 corrections as in pmc_traffic.py (KiB units, FETCH_SIZE x 2).  usage: dag_pmc_totals.py <fetch_dir> <write_dir> <executions> [label]"""
import csv, glob, os, sys


def total(d, counter):
    s, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                s += float(r["Counter_Value"]) * 1024.0
                n += 1
    return s, n


fetch, nf = total(sys.argv[1], "FETCH_SIZE")
write, nw = total(sys.argv[2], "WRITE_SIZE")
execs = int(sys.argv[3])
label = sys.argv[4] if len(sys.argv) > 4 else ""
print(f"{label}: {nf} / {nw} dispatches counted; per execution: fetch {2 * fetch / execs / 1e6:.1f} MB (x2 corrected), "
      f"write {write / execs / 1e6:.1f} MB, total {(2 * fetch + write) / execs / 1e6:.1f} MB")
