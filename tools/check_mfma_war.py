#!/usr/bin/env python
"""Build-time scan for early writes of MFMA source operands (round 6, NOTEBOOK.md, conv_wino3.hip):

    v_mfma_f32_32x32x16_bf16 v[96:111], v[176:179], v[196:199], v[96:111]     <- the last MFMA of an accumulate chain: it waits for its predecessor
    ds_read_b128 ...  (x3)
    v_cndmask_b32_e64 v196, 0, 1, s[74:75]                                    <- a VALU write of its B operand three instructions later

The B operand was dead behind the chain and the register allocator handed v196 to a temporary.  The compiler's hazard recogniser knows
write-after-read on SrcC only, the ISA manual lists no wait states for SrcA / SrcB -- and nothing says when an MFMA that is issued behind an
unfinished predecessor of its own chain reads its 128-bit operands (four VGPRs per lane: several read cycles).  Found while hunting a run-dependent
error of conv_wino3; closing it did NOT cure that error (8 of 24 fresh processes still wrong), so this is a PRECAUTION, not a measured hazard: the
kernel keeps such writes out of the window, and this script refuses a build in which a VALU instruction writes a VGPR of SrcA / SrcB of such an MFMA
within WINDOW issue slots behind it (LONE slots when no other MFMA was issued shortly before it: it starts at once) unless TWO
later MFMAs have been issued in between (the second of which, in the kernels here, waits for the first: the flagged MFMA has then started);
LDS / VMEM loads into such a register are reported as notes (their data arrives later than any chain lasts).
Scanned: the MFMAs with 128-bit source operands (four VGPRs per lane: the 16-deep bf16 form above); --all adds the float32 forms, whose
one-register sources the kernels of the default path overwrite at once, bit-exactly (conv_nm: 0 slots behind; parity tests).
    python tools/check_mfma_war.py [--report] [--all] <file.o|file.so> ...      (--report: print, never fail)"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_store_hazard as ch

WINDOW = 48      # issue slots scanned behind an MFMA that may be queued (another MFMA within CHAINED slots in front of it) ...
LONE = 16        # ... and behind one that starts at once (nothing in the matrix unit in front of it: conv_wino2's bias MFMA behind the tile's barrier)
CHAINED = 12
ALL = False      # --all: MFMAs with narrow source operands too
SAFE_MFMAS = 2   # later MFMAs behind which the scan of a path stops


def loads_into(mn, ops):
    if mn.startswith(("ds_read", "ds_load", "buffer_load", "global_load", "flat_load", "scratch_load")) and ops:
        r = ch.vrange(ops[0])
        return [r] if r else []
    return []


def check(funcs, where, notes):
    bad = []
    for fn, ins in funcs.items():
        by_addr = {a: i for i, (a, _, _) in enumerate(ins) if a is not None}
        for i, (addr, mn, ops) in enumerate(ins):
            if not mn.startswith("v_mfma") or len(ops) < 3:
                continue
            srcs = [r for r in (ch.vrange(ops[1]), ch.vrange(ops[2])) if r]
            # the MFMAs with 128-bit source operands (gfx950's 16-deep bf16 / f16 and 32-deep fp8 forms: four VGPRs per lane and operand) -- the
            # float32 forms with one-VGPR sources (conv_wino2, conv_nm, attention) run the same pattern bit-exactly (their parity tests; --all scans them too)
            if not ALL:
                srcs = [r for r in srcs if r[1] - r[0] >= 3]
            if not srcs:
                continue
            window = WINDOW if any(m.startswith("v_mfma") for (_, m, _) in ins[max(0, i - CHAINED):i]) else LONE
            work, seen, flagged = [(i + 1, 0, 0)], set(), False
            while work and not flagged:
                j, used, later = work.pop()
                if j >= len(ins) or used >= window or later >= SAFE_MFMAS or (j, used, later) in seen:
                    continue
                seen.add((j, used, later))
                _, m2, o2 = ins[j]
                if m2 == "s_endpgm" or m2 == "s_barrier":
                    continue
                if m2.startswith("v_mfma"):
                    work.append((j + 1, used + 1, later + 1))
                    continue
                hit = [r for r in ch.written(m2, o2) for s in srcs if not (r[1] < s[0] or r[0] > s[1])]
                if hit:
                    bad.append("%s: %s: `%s %s` is followed after %d slot(s) and %d MFMA(s) by `%s %s`" % (
                        where, fn[:70], mn, ", ".join(ops), used, later, m2, ", ".join(o2)))
                    flagged = True
                    break
                lh = [r for r in loads_into(m2, o2) for s in srcs if not (r[1] < s[0] or r[0] > s[1])]
                if lh:
                    notes.append("%s: %s: `%s %s` -> load `%s %s` after %d slot(s), %d MFMA(s)" % (where, fn[:70], mn, ", ".join(ops), m2, ", ".join(o2), used, later))
                    continue          # (the register is the load's from here on)
                if m2.startswith("s_cbranch") or m2 == "s_branch":
                    tgt = [o for o in o2 if o.startswith("@")]
                    base = ins[0][0]
                    if tgt and base is not None and base + int(tgt[0][1:]) in by_addr:
                        work.append((by_addr[base + int(tgt[0][1:])], used + 1, later))
                    if m2 == "s_branch":
                        continue
                work.append((j + 1, used + ch.slots(m2, o2), later))
    return bad


def main(argv):
    global ALL
    report = "--report" in argv
    ALL = "--all" in argv
    paths = [a for a in argv if not a.startswith("--")]
    tmp = tempfile.mkdtemp(prefix="mfmawar")
    bad, notes, nobj, nmfma = [], [], 0, 0
    try:
        for p in paths:
            local = os.path.join(tmp, os.path.basename(p))
            shutil.copy(p, local)
            subprocess.run([ch.OBJDUMP, "--offloading", local], capture_output=True, text=True)
            for co in sorted(glob.glob(local + ".*gfx950*")):
                funcs = ch.parse(co)
                nobj += 1
                nmfma += sum(1 for ins in funcs.values() for (_, mn, _) in ins if mn.startswith("v_mfma"))
                bad += check(funcs, os.path.basename(p), notes)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for b in bad:
        print("MFMA SOURCE WAR  " + b)
    if report:
        for n in notes[:12]:
            print("note  " + n)
    print("check_mfma_war: %d code object(s), %d MFMAs, %d VALU write(s) of a source operand inside the window; %d load(s) into one (noted)" % (nobj, nmfma, len(bad), len(notes)))
    if not nobj:
        print("check_mfma_war: no device code found -- refusing")
        return 1
    return 1 if bad and not report else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
