"""Instruction mix of the MFMA-heavy basic blocks of one kernel in a hipcc -S listing.
usage: python tools/asm_blocks.py /tmp/fused.s <mangled-name-substring> [min_mfma] [--hist label]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read(); key = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
hist = sys.argv[5] if len(sys.argv) > 5 else None
m = re.search(r"\n(_Z\w*" + re.escape(key) + r"\w*):", s)
body = s[m.end():]; body = body[:body.index("s_endpgm")]
blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
print(m.group(1))
for i in range(1, len(blocks), 2):
    lab, b = blocks[i], blocks[i + 1]
    ops = [l.split()[0] for l in b.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    c = Counter()
    for op in ops:
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["ds"] += 1
        elif op.startswith(("global_", "buffer_")): c["vmem"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    if c["mfma"] >= mn:
        print(lab, dict(c), len(ops))
    if hist and lab == hist:
        print(Counter(o for o in ops if o.startswith(("v_", "ds_", "scratch"))).most_common(40))
