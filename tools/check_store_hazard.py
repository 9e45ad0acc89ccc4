#!/usr/bin/env python3
"""Static check of the product kernels' ISA for the hazard found in round 3 (profiles/r03/experiments.md section 6): a 12- or
16-byte buffer / global store whose data registers are overwritten by a VALU instruction within the next two issue slots.
The compiler's hazard table covers the case "soffset is not an SGPR" only; on MI355X the first launch of a process showed the
SGPR-soffset case corrupting data as well.  Scans the .s files `make -C gpu_video_codec_amd/csrc asm`-style compiles produce:
    python3 tools/check_store_hazard.py [file.s ...]     (default: compiles the three .hip files of the product to /tmp)
Exit code 1 if a suspicious sequence is found."""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu_video_codec_amd", "csrc")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    bad = []
    kernel = "?"
    lines = open(path).read().split("\n")
    ins = []
    for ln in lines:
        t = ln.strip()
        if t.endswith(":") and t.startswith("_Z"):
            kernel = t[:-1]
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        ins.append((kernel, t))
    for i, (k, t) in enumerate(ins):
        m = re.match(r"(buffer|global|flat|scratch)_store_dwordx([34])\s+(\S+?),", t)
        if not m:
            continue
        data = regs(m.group(3)) if m.group(1) == "buffer" else set()
        if m.group(1) != "buffer":  # global_store vaddr, vdata, ...
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            data = regs(ops[1]) if len(ops) > 1 else set()
        slots = 0
        for k2, t2 in ins[i + 1:i + 6]:
            if k2 != k:
                break
            op = t2.split()[0]
            if op == "s_nop":
                slots += 1 + int(t2.split()[1])
                continue
            if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readlane"):
                dst = t2.split(None, 1)[1].split(",")[0].strip()
                if slots < 2 and regs(dst) & data:
                    bad.append((os.path.basename(path), k, t, t2, slots))
                    break
            slots += 1
            if slots >= 2:
                break
    return bad


def main():
    files = sys.argv[1:]
    if not files:
        for src in ("deblock_kernels.hip", "deblock_h265.hip", "sao.hip"):
            out = "/tmp/hazard_%s.s" % src.split(".")[0]
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                                   os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
            files.append(out)
    bad = []
    for f in files:
        bad += scan(f)
    for b in bad:
        print("HAZARD? %s  %s\n    %s\n    %s   (%d wait states between)" % b)
    print("%d suspicious store / overwrite sequence(s) in %d file(s)" % (len(bad), len(files)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
