#!/usr/bin/env python3
"""Development aid: one kernel out of a `hipcc -S --cuda-device-only` listing, with its loop spans, mnemonic histogram and the
resource comment block.  usage: isa_fn.py listing.s <substring of the mangled name> [--dump out.s]"""
import re, sys
from collections import Counter
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
body = src[start:end]
meta = [l for l in src[end:end + 40] if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumSgprs", l)]
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
lab = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB[0-9_]+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.match(r"^\t(s_cbranch_\w+|s_branch)\s+(\.LBB[0-9_]+)", l)
    if m and m.group(2) in lab and lab[m.group(2)] < i:
        loops.append((lab[m.group(2)], i))
c = Counter(m.group(1) for l in body if (m := re.match(r"^\t([a-z_0-9]+)", l)))
print(src[start][:100])
print("\n".join(meta))
print("lines", len(body), "loops (start, end, span):", [(a, b, b - a) for a, b in sorted(loops, key=lambda x: x[0] - x[1])[:6]])
print(c.most_common(30))
