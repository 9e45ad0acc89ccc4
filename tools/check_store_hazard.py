#!/usr/bin/env python3
"""Static check of SHIPPED gfx950 machine code for the store hazard found in round 3 (profiles/r04/store_hazard.md): a 12- or
16-byte buffer / global / flat / scratch store whose data registers a VALU instruction overwrites within the next two wait
states.  The compiler's hazard table inserts `s_nop 1` on gfx940-family targets for such stores EXCEPT buffer stores whose
soffset is an SGPR; on MI355X exactly that exempted form corrupted data (the first launch of a process wrote the next
ds_read address into a few hundred samples per 4K frame).

What is scanned is what ships: the gfx950 code objects are pulled out of the offload bundles inside shared libraries, executables
and object files, disassembled with llvm-objdump, and every wide store is followed along BOTH sides of every branch (a store
that ends a basic block is checked against the first instructions of each successor).  Compiler `.s` files are accepted too.

    python3 tools/check_store_hazard.py [file ...]
        default: gpu_video_codec_amd/libhevcdbk.so, libhevcdbk_diag.so and the built tools/ubench programs
Exit code 1 if a suspicious sequence is found (run by `make all` in gpu_video_codec_amd/csrc and tools/ubench: fails the build).
--self-test: the scanner against hand-written sequences (no toolchain needed)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WAIT_STATES = 2  # what the compiler itself inserts on gfx942 / gfx950 for the forms its table covers (one on gfx90a)


def regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def operands(text):
    parts = text.split(None, 1)
    return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []


def store_data(text):
    """VGPRs holding the data of a store of more than 8 bytes, or None"""
    m = re.match(r"(buffer|global|flat|scratch)_store_(dwordx[34]|b96|b128)\b", text)
    if not m:
        return None
    ops = operands(text)
    if not ops:
        return None
    return regs(ops[0]) if m.group(1) == "buffer" else (regs(ops[1]) if len(ops) > 1 else set())


def valu_writes(text):
    """VGPRs a VALU instruction writes (empty for compares, lane reads and anything that is not VALU)"""
    op = text.split()[0]
    if not op.startswith("v_") or op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop", "v_accvgpr_write")):
        return set()
    ops = operands(text)
    if not ops:
        return set()
    out = regs(ops[0])
    if op.startswith("v_swap") and len(ops) > 1:
        out |= regs(ops[1])
    return out


def parse(lines):
    """-> (instructions [(function, text)], labels {(function, label): index of the next instruction})"""
    ins, labels, func = [], {}, "?"
    for ln in lines:
        t = ln.split("//")[0].split(";")[0].rstrip()
        s = t.strip()
        if not s:
            continue
        m = re.match(r"^[0-9a-f]{8,16} <([^>]+)>:$", s) or re.match(r"^([A-Za-z_.$][\w.$]*):$", s)
        if m:
            name = m.group(1)
            if re.match(r"^(L\d+|\.LBB\S*|\.Ltmp\S*|\.LFB\S*)$", name):
                labels[(func, name)] = len(ins)
            elif not name.startswith("."):
                func = name
            continue
        if s.startswith(".") or not ln[:1].isspace():
            continue
        ins.append((func, s))
    return ins, labels


def scan_stream(ins, labels, origin):
    bad = []
    for i, (func, text) in enumerate(ins):
        data = store_data(text)
        if not data:
            continue
        # every path out of the store, until WAIT_STATES wait states have passed
        todo, seen = [(i + 1, 0)], set()
        while todo:
            j, slots = todo.pop()
            while slots < WAIT_STATES and j < len(ins) and ins[j][0] == func and (j, slots) not in seen:
                seen.add((j, slots))
                t2 = ins[j][1]
                op = t2.split()[0]
                if op == "s_endpgm":
                    break
                if op == "s_nop":
                    slots += 1 + int(operands(t2)[0], 0)
                    j += 1
                    continue
                if op in ("s_branch",) or op.startswith("s_cbranch"):
                    # the branch itself is one issued instruction = one wait state, on both sides (the compiler's hazard
                    # recognizer counts it the same way when it looks back across a block boundary)
                    slots += 1
                    tgt = labels.get((func, operands(t2)[0]))
                    if tgt is not None:
                        todo.append((tgt, slots))
                    if op == "s_branch":
                        break
                    j += 1
                    continue
                hit = valu_writes(t2) & data
                if hit:
                    bad.append((origin, func, text, t2, slots))
                    todo = []
                    break
                slots += 1
                j += 1
    return bad


def code_objects(path):
    """gfx950 code objects inside the clang offload bundles of a binary"""
    blob = open(path, "rb").read()
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        for _ in range(min(n, 64)):
            o, sz, ts = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + ts].decode(errors="replace")
            off += ts
            if "gfx950" in triple and sz:
                out.append(blob[i + o:i + o + sz])
        pos = i + 24


def scan_file(path):
    if path.endswith(".s"):
        ins, labels = parse(open(path).read().split("\n"))
        return scan_stream(ins, labels, os.path.basename(path)), 1
    bad, n = [], 0
    for k, co in enumerate(code_objects(path)):
        with tempfile.NamedTemporaryFile(suffix=".elf") as tmp:
            tmp.write(co)
            tmp.flush()
            txt = subprocess.check_output([OBJDUMP, "-d", "--symbolize-operands", tmp.name], stderr=subprocess.DEVNULL).decode(errors="replace")
        ins, labels = parse(txt.split("\n"))
        bad += scan_stream(ins, labels, "%s[code object %d]" % (os.path.basename(path), k))
        n += 1
    return bad, n


SELF_TEST = [
    # (expected number of findings, listing)
    (1, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\tv_add_u32_e32 v2, v4, v1\n\ts_endpgm\n"),
    (1, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\ts_nop 0\n\tv_add_u32_e32 v3, v4, v1\n\ts_endpgm\n"),
    (0, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\ts_nop 1\n\tv_add_u32_e32 v2, v4, v1\n\ts_endpgm\n"),
    (0, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\ts_mov_b32 s1, 0\n\ts_mov_b32 s2, 0\n\tv_add_u32_e32 v2, v4, v1\n"),
    (0, "f:\n\tbuffer_store_dwordx2 v[2:3], v6, s[4:7], s8 offen\n\tv_add_u32_e32 v2, v4, v1\n\ts_endpgm\n"),
    (1, "f:\n\tglobal_store_dwordx4 v[8:9], v[2:5], off\n\tv_mov_b32_e32 v5, 0\n\ts_endpgm\n"),
    (0, "f:\n\tglobal_store_dwordx4 v[8:9], v[2:5], off\n\tv_mov_b32_e32 v8, 0\n\ts_endpgm\n"),
    # the store ends a block: the overwrite sits at the branch target / behind the label
    (1, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\ts_cbranch_execz .LBB0_2\n\ts_nop 1\n\tv_mov_b32_e32 v9, 0\n.LBB0_2:\n\tv_mov_b32_e32 v4, 0\n\ts_endpgm\n"),
    (0, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\ts_mov_b32 s1, 0\n\ts_cbranch_execz .LBB0_2\n\ts_nop 1\n.LBB0_2:\n\tv_mov_b32_e32 v4, 0\n\ts_endpgm\n"),
    (1, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\ts_branch .LBB0_3\n.LBB0_2:\n\ts_nop 4\n.LBB0_3:\n\tv_mov_b32_e32 v4, 0\n\ts_endpgm\n"),
    (0, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\ts_cbranch_execz .LBB0_2\n\ts_nop 0\n\tv_mov_b32_e32 v4, 0\n.LBB0_2:\n\ts_nop 0\n\tv_mov_b32_e32 v4, 0\n\ts_endpgm\n"),
    (1, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\tv_swap_b32 v9, v3\n\ts_endpgm\n"),
    (0, "f:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen\n\tv_cmp_eq_u32_e32 vcc, v2, v3\n\tv_readfirstlane_b32 s3, v2\n\tv_mov_b32_e32 v2, 0\n"),
    # objdump form: addresses, symbolized labels, trailing encodings
    (1, "0000000000001000 <_Z1kv>:\n\tbuffer_store_dwordx4 v[2:5], v6, s[4:7], s8 offen        // 000000001000: E07C1000\n"
        "\ts_cbranch_execz L0                                    // 000000001008: BF880002\n\ts_nop 7\n0000000000001010 <L0>:\n"
        "\tv_mov_b32_e32 v3, 0                                   // 000000001010: 7E060280\n"),
]


def self_test():
    ok = True
    for want, text in SELF_TEST:
        ins, labels = parse(text.split("\n"))
        got = len(scan_stream(ins, labels, "self-test"))
        if got != want:
            ok = False
            print("SELF-TEST FAILED: expected %d finding(s), got %d in\n%s" % (want, got, text))
    print("self-test %s (%d sequences)" % ("passed" if ok else "FAILED", len(SELF_TEST)))
    return 0 if ok else 1


def main():
    args = sys.argv[1:]
    if args == ["--self-test"]:
        return self_test()
    files = args
    if not files:
        cand = [os.path.join(ROOT, "gpu_video_codec_amd", n) for n in ("libhevcdbk.so", "libhevcdbk_diag.so")]
        ub = os.path.join(ROOT, "tools", "ubench")
        cand += [os.path.join(ub, n) for n in ("valu_rate", "valu_rate2", "valu_rate3", "copy_bw", "pcie_bw", "host_stage", "bar_write")]
        files = [f for f in cand if os.path.exists(f)]
        if not any(f.endswith("libhevcdbk.so") for f in files):
            print("libhevcdbk.so has not been built: nothing to scan")
            return 1
    bad, n_obj = [], 0
    for f in files:
        b, n = scan_file(f)
        bad += b
        n_obj += n
    for b in bad:
        print("HAZARD? %s  %s\n    %s\n    %s   (%d wait state(s) between)" % b)
    print("%d suspicious store / overwrite sequence(s) in %d code object(s) of %d file(s)" % (len(bad), n_obj, len(files)))
    return 1 if bad or n_obj == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
