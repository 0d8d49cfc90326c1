#!/usr/bin/env python
"""Build-time scan of the device code for the store-data hazard found in round 5 (NOTEBOOK.md):

    buffer_store_dwordx4 v[72:75], v104, s[44:47], s76 offen
    v_cndmask_b32_e64 v72, 0, 1, s[58:59]        <- a VALU write of the store's data register in the next slot

On gfx950 the store then writes the NEW value for the lanes it had not read yet (the last quad of each 16-lane group).  The
ISA manual asks for one wait state between a VMEM store of more than 64 bits and a write of its data VGPRs but exempts MUBUF
stores whose soffset is an SGPR; the compiler's hazard recogniser follows the manual; the hardware does not.  This script
disassembles every gfx950 code object of the given files (.o / .so: llvm-objdump --offloading) and fails if a VALU
instruction writes a data register of a > 64-bit store in the very next issue slot (measured with
tools/experiments/pw_repro.hip: no slot in between = wrong values in 721 of 1500 launches, ONE wait state = 0 of 3000; s_nop N
counts N + 1 slots; both arms of a branch are followed); writes one slot later are counted and reported, not refused.
    python tools/check_store_hazard.py <file.o|file.so> ..."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile



def find_objdump():
    """llvm-objdump of the toolchain that built the objects: $OBJDUMP, else next to $HIPCC's clang, else $ROCM_PATH, else /opt/rocm."""
    cands = [os.environ.get("OBJDUMP")]
    hipcc = shutil.which(os.environ.get("HIPCC", "hipcc"))
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-objdump"), shutil.which("llvm-objdump")]
    for c in cands:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    sys.exit("check_store_hazard: llvm-objdump not found (set OBJDUMP=/path/to/llvm-objdump; looked next to $HIPCC and under $ROCM_PATH)")


OBJDUMP = find_objdump()
STORE = re.compile(r"^(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\b")
VREG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")
NEED = 1                       # issue slots the hardware needs between the store and the write
LOOK = 2                       # ... and how far the scan looks (writes at distance >= NEED are reported as notes)


def vrange(tok):
    m = VREG.match(tok.strip())
    if not m:
        return None
    if m.group(1) is not None:
        return int(m.group(1)), int(m.group(1))
    return int(m.group(2)), int(m.group(3))


def parse(path):
    """-> {function: [(addr, mnemonic, [operands])]}"""
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for ln in out.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            continue
        if cur is None or not ln.startswith(" ") and not ln.startswith("\t"):
            continue
        body = ln.split("//")[0].strip()
        if not body:
            continue
        am = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
        addr = int(am.group(1), 16) if am else None
        parts = body.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        tm = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$", ln)              # a branch's target: <function+0xOFFSET> in the comment
        if tm and (parts[0].startswith("s_cbranch") or parts[0] == "s_branch"):
            ops = ops + ["@%d" % int(tm.group(1), 16)]
        cur.append((addr, parts[0], ops))
    return funcs


def written(mn, ops):
    """VGPR ranges a VALU-class instruction writes ([]: none / not a VALU instruction).  v_swap_b32 and v_permlane16/32_swap_b32
    write BOTH their operands; a VOP3 form whose first destination is an SGPR pair (v_add_co_u32 v1, vcc, ..: the VGPR comes first;
    v_cmp / v_readlane: no VGPR destination at all) is covered by taking the first operands that parse as VGPRs."""
    if not mn.startswith("v_") or mn.startswith("v_cmp") and not mn.startswith("v_cmpx") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane"):
        return []
    if not ops:
        return []
    if "swap" in mn:
        return [r for r in (vrange(o) for o in ops[:2]) if r]
    r = vrange(ops[0])
    return [r] if r else []


def slots(mn, ops):
    if mn == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def check(funcs, where, notes):
    bad = []
    for fn, ins in funcs.items():
        by_addr = {a: i for i, (a, _, _) in enumerate(ins) if a is not None}
        for i, (addr, mn, ops) in enumerate(ins):
            if not STORE.match(mn):
                continue
            data = vrange(ops[0]) if mn.startswith("buffer") else (vrange(ops[1]) if len(ops) > 1 else None)
            if data is None:
                continue
            # walk the next NEED slots along every path
            work = [(i + 1, 0)]
            seen = set()
            while work:
                j, used = work.pop()
                if j >= len(ins) or used >= LOOK or (j, used) in seen:
                    continue
                seen.add((j, used))
                _, m2, o2 = ins[j]
                w = [r for r in written(m2, o2) if not (r[1] < data[0] or r[0] > data[1])]
                if w:
                    (bad if used < NEED else notes).append("%s: %s: `%s %s` is followed after %d slot(s) by `%s %s`" % (
                        where, fn[:80], mn, ", ".join(ops), used, m2, ", ".join(o2)))
                    break
                if m2 in ("s_endpgm",):
                    continue
                if m2.startswith("s_cbranch") or m2 == "s_branch":
                    tgt = [o for o in o2 if o.startswith("@")]
                    base = ins[0][0]
                    if tgt and base is not None and base + int(tgt[0][1:]) in by_addr:
                        work.append((by_addr[base + int(tgt[0][1:])], used + 1))
                    elif m2 == "s_branch":
                        bad.append("%s: %s: unresolved branch behind a wide store (scanner)" % (where, fn[:80]))
                    if m2 == "s_branch":
                        continue
                work.append((j + 1, used + slots(m2, o2)))
    return bad


def main(paths):
    tmp = tempfile.mkdtemp(prefix="storehaz")
    bad, notes, nobj, nstores, empty = [], [], 0, 0, []
    try:
        for p in paths:
            local = os.path.join(tmp, os.path.basename(p))
            shutil.copy(p, local)
            r = subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True)
            cos = sorted(glob.glob(local + ".*gfx950*"))
            if not cos:          # (host-only objects have none; reported only if NOTHING was found -- then the build is refused below)
                empty.append("%s: `%s --offloading` extracted no gfx950 code object (rc %d) %s" % (p, OBJDUMP, r.returncode, r.stderr[-300:]))
            for co in cos:
                funcs = parse(co)
                nobj += 1
                nstores += sum(1 for ins in funcs.values() for (_, mn, _) in ins if STORE.match(mn))
                bad += check(funcs, os.path.basename(p), notes)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for b in bad:
        print("STORE-DATA HAZARD  " + b)
    if not nobj:
        print("check_store_hazard: no device code found -- refusing\n  " + "\n  ".join(empty))
    print("check_store_hazard: %d code object(s), %d stores of more than 64 bits, %d hazard(s); %d write(s) one slot later (safe: noted)" % (
        nobj, nstores, len(bad), len(notes)))
    return 1 if bad or not nobj else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
