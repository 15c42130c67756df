#!/usr/bin/env python
"""Instruction mix of one kernel in a hipcc -S dump:  isa_mix.py file.s <substring of mangled name>"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start + 1, len(src)) if src[i].startswith("\t.section") or src[i].startswith(".Lfunc_end"))
ops = collections.Counter()
for l in src[start:end]:
    m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)", l)
    if m:
        ops[m.group(1)] += 1
valu = sum(v for k, v in ops.items() if k.startswith("v_"))
mul = sum(v for k, v in ops.items() if re.match(r"v_(mul|mad)_", k))
print(f"{src[start][:100]}\n  VALU {valu}  (multiplies {mul})  SALU {sum(v for k,v in ops.items() if k.startswith('s_'))}  "
      f"LDS {sum(v for k,v in ops.items() if k.startswith('ds_'))}  VMEM {sum(v for k,v in ops.items() if k.startswith(('global','buffer','flat')))}")
if len(sys.argv) > 3:
    for k, v in ops.most_common(25):
        print(f"   {v:5d} {k}")
