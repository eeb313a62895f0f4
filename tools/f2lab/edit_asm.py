"""Line edits of a device listing.  usage: edit_asm.py <in.s> <out.s> <op> ...   with op = after:<line>:<file-or-text> | before:<line>:<text> | replace:<line>:<text> | delete:<line>
Text uses ';' as the instruction separator; line numbers refer to the INPUT file (1-based)."""
import sys
src, dst, ops = sys.argv[1], sys.argv[2], sys.argv[3:]
L = open(src).read().split("\n")
before, after, repl, dele = {}, {}, {}, set()
def ins(t):
    return ["\t" + x.strip() for x in t.split(";") if x.strip()]
for op in ops:
    kind, line, *rest = op.split(":", 2)
    line = int(line); text = rest[0] if rest else ""
    if kind == "after": after.setdefault(line, []).extend(ins(text))
    elif kind == "before": before.setdefault(line, []).extend(ins(text))
    elif kind == "replace": repl[line] = ins(text)
    elif kind == "delete": dele.add(line)
    else: raise SystemExit("bad op " + op)
out = []
for i, l in enumerate(L, start=1):
    out += before.get(i, [])
    if i in repl: out += repl[i]
    elif i not in dele: out.append(l)
    out += after.get(i, [])
open(dst, "w").write("\n".join(out))
print("edited", len(ops), "ops ->", dst)
