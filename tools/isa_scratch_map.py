"""Where the scratch (spill) accesses of one kernel sit relative to its loops: basic-block map of a hipcc -S listing.
For every basic block: its label, the loop nest the compiler annotated ("in Loop: Header=... Depth=n"), instruction count, and the
scratch / global / LDS / MFMA instructions it holds.  Shows, for march_accel_kernel<9,256,0>, that the blocks of the march step and
the colour evaluation (loop depth >= 1, between the LDS grid read and the row loads' consumers) hold no scratch access.
usage: isa_scratch_map.py file.s mangled-name-substring"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], {"label": "entry", "loop": "", "n": 0, "scratch": 0, "global_load": 0, "global_store": 0, "lds": 0, "f64": 0, "line": start}
for i in range(start + 1, end + 1):
    t = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append(cur)
        loop = re.search(r"(in Loop: Header=\S+ Depth=\d+|Loop Header: Depth=\d+)", lines[i])
        cur = {"label": m.group(1), "loop": loop.group(1) if loop else "", "n": 0, "scratch": 0, "global_load": 0, "global_store": 0, "lds": 0, "f64": 0, "line": i}
        continue
    if not t or t.startswith((";", ".", "//")):
        if "Loop" in t and not cur["loop"]:
            loop = re.search(r"(in Loop: Header=\S+ Depth=\d+|Loop Header: Depth=\d+|Inner Loop Header: Depth=\d+)", t)
            if loop:
                cur["loop"] = loop.group(1)
        continue
    op = t.split()[0]
    cur["n"] += 1
    if op.startswith("scratch_"): cur["scratch"] += 1
    elif op.startswith("global_load"): cur["global_load"] += 1
    elif op.startswith(("global_store", "global_atomic")): cur["global_store"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    if "_f64" in op: cur["f64"] += 1
blocks.append(cur)
tot = sum(b["scratch"] for b in blocks)
in_loop = sum(b["scratch"] for b in blocks if b["loop"])
print(f"kernel {key}: {sum(b['n'] for b in blocks)} instructions in {len(blocks)} blocks, {tot} scratch accesses, {in_loop} of them in blocks the compiler places inside a loop")
print(f"{'block':14s} {'instr':>5s} {'scratch':>7s} {'gload':>5s} {'gstore':>6s} {'lds':>4s} {'f64':>4s}  loop")
for b in blocks:
    if b["n"]:
        print(f"{b['label']:14s} {b['n']:5d} {b['scratch']:7d} {b['global_load']:5d} {b['global_store']:6d} {b['lds']:4d} {b['f64']:4d}  {b['loop']}")
