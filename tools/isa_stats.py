#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -S dump (make -C gpu_video_codec_amd/csrc asm)."""
import re
import sys
from collections import Counter

path = sys.argv[1] if len(sys.argv) > 1 else "gpu_video_codec_amd/csrc/deblock_kernels.s"
pat = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path).read().split("\n")
cur, body = None, {}
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
        body[cur] = []
        continue
    if cur and l.startswith("\t.amdhsa_kernel") or l.startswith(".Lfunc_end"):
        cur = None
    if cur and l.startswith("\t") and not l.strip().startswith((".", ";")):
        body[cur].append(l.strip().split()[0])
for name, ins in body.items():
    if pat not in name:
        continue
    c = Counter(ins)
    tot = lambda p: sum(n for k, n in c.items() if k.startswith(p))
    print("%s\n  total %d  valu %d (v_pk %d, v_perm %d, cndmask %d)  salu %d  vmem %d  branch %d" % (
        name, len(ins), tot("v_"), tot("v_pk"), c.get("v_perm_b32", 0), tot("v_cndmask"), tot("s_"),
        tot("global_") + tot("buffer_") + tot("flat_"), tot("s_cbranch")))
    if len(sys.argv) > 3:
        print("  ", sorted(c.items(), key=lambda x: -x[1])[:45])
