"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports wide coalesced reads by 2x
-> doubled here; both counters are in KiB.  usage: pmc_traffic.py <fetch_dir> <write_dir> [out.json]"""
import collections, csv, glob, json, os, sys

CLASS = [("ks_inner_kernel", "ks_mac"), ("k_ks_mac", "ks_mac"), ("OpKsDigit", "ksdigit"), ("OpModDown", "moddown"),
         ("OpRRLast", "intt"), ("OpRR", "moddown"), ("k_mul22", "elementwise"), ("k_square", "elementwise"),
         ("k_addsub", "elementwise"), ("k_mul_plain", "elementwise"), ("k_galois", "elementwise")]


def classify(name):
    base = None
    for key, cls in CLASS:
        if key in name:
            base = cls
            break
    if "ntt_loop_kernel" in name:  # the looped contiguous pass: inverse = first pass of an INTT, forward = second pass
        import re
        m = re.search(r"ntt_loop_kernel<\d+, \d+, (true|false)", name)
        if m.group(1) == "true":
            return "intt_pass1"
        return f"{base or 'ntt'}_pass2"
    if "ntt_pass_kernel" in name:
        import re
        m = re.search(r"ntt_pass_kernel<\d+, \d+, (true|false), (true|false)", name)
        strided, inverse = m.group(1) == "true", m.group(2) == "true"
        if inverse:
            return "intt_pass2" if strided else "intt_pass1"
        base = base or "ntt"
        return f"{base}_pass1" if strided else f"{base}_pass2"
    return base or name[:40]


KERNELS = set()


def load(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[classify(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
                KERNELS.add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    return out


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    f = 2.0 * sum(fetch[k]) / max(len(fetch[k]), 1)
    w = sum(write[k]) / max(len(write[k]), 1)
    res[k] = {"launches": max(len(fetch[k]), len(write[k])), "fetch_bytes_per_launch": round(f), "write_bytes_per_launch": round(w),
              "hbm_bytes_per_launch": round(f + w)}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_amd.roofline import csrc_tree_hash  # noqa: E402
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py; FETCH_SIZE x2 (gfx950)",
       "tree": csrc_tree_hash(), "commit": os.environ.get("EVA_COMMIT", ""), "kernel_names": sorted(KERNELS), "by_class": res}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
