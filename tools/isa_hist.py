"""Instruction histogram of one kernel in a hipcc -save-temps .s file.
usage: isa_hist.py file.s mangled-name-substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
lines = s.split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start + 1:end]
ins = []
for l in body:
    t = l.strip()
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
fam = collections.Counter()
for k, v in c.items():
    f = ("f64" if "_f64" in k else "valu") if k.startswith("v_") else "salu" if k.startswith("s_") else \
        "vmem" if k.startswith(("global_", "buffer_", "flat_")) else "lds" if k.startswith("ds_") else "other"
    fam[f] += v
print(len(ins), "instructions", dict(fam))
print(c.most_common(45))
i = s.find(".name:", s.find(key, s.find(".amdhsa_kernel")))
for m in re.finditer(r"\.name:\s+(\S*%s\S*)" % re.escape(key), s):
    blk = s[max(0, m.start() - 1500):m.start() + 1500]
    print(m.group(1)[:60], sorted(set(re.findall(r"\.(vgpr_count|sgpr_count|agpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\d+)", blk))))
