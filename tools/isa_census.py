#!/usr/bin/env python
"""Basic-block census of a kernel's gfx950 code: per block the MFMA, SGPR-spill lane moves (v_readlane / v_writelane), scratch,
v_cndmask, vector-memory, LDS, other-VALU and SALU counts, with the branch edges -- what the spill gate of csrc/Makefile and
NOTEBOOK.md's instruction counts are read from.
    python tools/isa_census.py <file.o> <kernel-name-substring> [--blocks] [--gate N]
--gate N: exit 1 if a block of >= 16 MFMAs of a matching kernel holds more than N lane moves or any scratch access (csrc/Makefile)."""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def disassemble(obj):
    """-> {function: [(addr, text, branch target or None)]} over every gfx950 code object bundled in obj"""
    tmp = tempfile.mkdtemp(prefix="isa_census_")
    base = os.path.join(tmp, os.path.basename(obj))
    with open(obj, "rb") as f, open(base, "wb") as g:
        g.write(f.read())
    subprocess.run([OBJDUMP, "--offloading", base], cwd=tmp, capture_output=True)
    funcs = {}
    for fn in sorted(os.listdir(tmp)):
        if "amdgcn" not in fn:
            continue
        out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, fn)], capture_output=True, text=True).stdout
        cur, start = None, 0
        for ln in out.splitlines():
            m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
            if m:
                cur, start = funcs.setdefault(m.group(2), []), int(m.group(1), 16)
                continue
            am = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
            if cur is None or not am:
                continue
            body = ln.split("//")[0].strip()
            tm = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", ln)
            cur.append((int(am.group(1), 16), body, start + int(tm.group(1), 16) if tm and "branch" in body else None))
    return funcs


def blocks_of(ins):
    targets = {t for _, _, t in ins if t is not None}
    blocks, cur = [], []
    for a, b, t in ins:
        if a in targets and cur:
            blocks.append(cur)
            cur = []
        cur.append((a, b, t))
        if "branch" in b or b.startswith("s_endpgm"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    return blocks


def stats(bl):
    c = lambda p: sum(1 for _, x, _ in bl if x.startswith(p))      # noqa: E731
    return dict(n=len(bl), mfma=c("v_mfma"), rdlane=c("v_readlane"), wrlane=c("v_writelane"), cnd=c("v_cndmask"), scratch=c("scratch_"),
                vmem=c("buffer_") + c("global_"), lds=c("ds_"),
                valu=sum(1 for _, x, _ in bl if x.startswith("v_") and not x.startswith("v_mfma")), salu=c("s_"))


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    gate = int(sys.argv[sys.argv.index("--gate") + 1]) if "--gate" in sys.argv else None
    bad, seen = [], 0
    for name, ins in disassemble(obj).items():
        if pat not in name:
            continue
        bls = blocks_of(ins)
        tot = stats(ins)
        print("%s: %d instructions, %d blocks | v_readlane %d v_writelane %d scratch %d v_cndmask %d" % (
            name, tot["n"], len(bls), tot["rdlane"], tot["wrlane"], tot["scratch"], tot["cnd"]))
        mf = [stats(b) for b in bls if stats(b)["mfma"] >= 16]
        seen += 1
        bad += ["%s: a %d-MFMA block with %d lane moves, %d scratch accesses" % (name, s["mfma"], s["rdlane"] + s["wrlane"], s["scratch"])
                for s in mf if gate is not None and (s["rdlane"] + s["wrlane"] > gate or s["scratch"])]
        print("   MFMA blocks: " + "; ".join("%d mfma + %d other valu (%d lane moves, %d scratch)" % (
            s["mfma"], s["valu"], s["rdlane"] + s["wrlane"], s["scratch"]) for s in mf))
        if "--blocks" in sys.argv:
            for b in bls:
                s = stats(b)
                a, x, t = b[-1]
                print("   %06x n=%3d mfma=%2d rl=%2d wl=%2d scr=%2d cnd=%2d vmem=%2d lds=%2d valu=%3d salu=%3d -> %s%s" % (
                    b[0][0], s["n"], s["mfma"], s["rdlane"], s["wrlane"], s["scratch"], s["cnd"], s["vmem"], s["lds"], s["valu"], s["salu"],
                    x.split()[0] if "branch" in x else "fall", " %06x" % t if t else ""))


    if gate is not None and (bad or not seen):
        print("isa_census gate (%d lane moves per MFMA block, no scratch): REFUSED\n  " % gate + "\n  ".join(bad or ["no kernel matched " + pat]))
        sys.exit(1)


if __name__ == "__main__":
    main()
