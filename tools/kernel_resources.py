#!/usr/bin/env python3
"""Register / LDS / scratch table of every kernel of the PRODUCT build, from `make -C gpu_video_codec_amd/csrc asm`
(hipcc -Rpass-analysis=kernel-resource-usage).   python3 tools/kernel_resources.py > profiles/<round>/kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "gpu_video_codec_amd", "csrc", "deblock_kernels.resources.txt")
txt = open(path).read()
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
print("# kernel resources of gpu_video_codec_amd/csrc/deblock_kernels.hip (product build, gfx950), commit %s or its working tree" % head)
print("%-6s %-6s %-8s %-10s %-6s %s" % ("VGPR", "SGPR", "scratch", "waves/SIMD", "LDS", "kernel"))
cur = {}
for line in txt.split("\n"):
    m = re.search(r"remark:\s+(.*)$", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip().split()[0]}
    for key, tag in (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), ("ScratchSize [bytes/lane]", "scratch"), ("Occupancy [waves/SIMD]", "occ"),
                     ("LDS Size [bytes/block]", "lds")):
        if t.startswith(key + ":"):
            cur[tag] = t.split(":", 1)[1].strip().split()[0]
    if "lds" in cur:
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip() or cur["name"]
        print("%-6s %-6s %-8s %-10s %-6s %s" % (cur.get("vgpr"), cur.get("sgpr"), cur.get("scratch"), cur.get("occ"), cur["lds"], name))
        cur = {}
