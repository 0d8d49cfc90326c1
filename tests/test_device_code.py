"""Static checks of the compiled device code (CPU: hipcc cross-compiles gfx950 without a GPU).

The store-data hazard of gfx950 found in round 5 (NOTEBOOK.md): a VALU write of a 16-byte buffer store's data register in the
next issue slot makes the store write the NEW value for the lanes it had not read yet.  The ISA manual exempts buffer stores
with an SGPR soffset from the one wait state it asks for, the compiler follows the manual, the hardware does not.
tools/check_store_hazard.py scans the disassembly; the build refuses a library with such a pair (csrc/Makefile)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tools", "check_store_hazard.py")
CSRC = os.path.join(ROOT, "ipdm-pytorch_amd", "csrc")


def test_no_store_data_hazard_in_the_built_libraries():
    objs = sorted(glob.glob(os.path.join(CSRC, "*.o")))
    assert len(objs) >= 14, "csrc/*.o missing: run __graft_entry__.build()"
    r = subprocess.run([sys.executable, CHECK] + objs, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 hazard(s)" in r.stdout, r.stdout[-500:]


def test_the_scanner_flags_the_shelved_pointwise_kernel(tmp_path):
    """Positive control: tools/experiments/conv_pw.hip is the round-4 kernel that produced run-dependent zeros (accumulator
    register 0, lanes 12-15 / 28-31) -- in its compiled code every instantiation has `buffer_store_dwordx4 v[a:a+3], .., sN offen`
    followed IMMEDIATELY by `v_cndmask_b32 v[a], 0, 1, ..` (tools/experiments/pw_repro.hip reproduces the zeros on the GPU:
    721 of 1500 launches; with one wait state behind the store: 0 of 3000)."""
    obj = str(tmp_path / "shelved_pw.o")
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-function", "-Wno-unused-value", "-I" + CSRC, "-c",
           os.path.join(ROOT, "tools", "experiments", "pw_repro.hip"), "-o", obj]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, CHECK, obj], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1, r.stdout[-2000:]
    flagged = [ln for ln in r.stdout.splitlines() if ln.startswith("STORE-DATA HAZARD")]
    assert len(flagged) >= 3 and all("after 0 slot(s)" in ln and "v_cndmask_b32" in ln for ln in flagged), r.stdout[-2000:]
    # ... and the same source with one wait state behind every 16-byte store is clean
    cmd2 = cmd[:-4] + ["-DPW_NOP_AFTER_STORE=0"] + cmd[-4:]
    cmd2[-1] = str(tmp_path / "shelved_pw_nop.o")
    subprocess.run(cmd2, check=True, capture_output=True, timeout=900)
    r2 = subprocess.run([sys.executable, CHECK, cmd2[-1]], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-2000:]


def test_the_scanner_treats_both_operands_of_a_swap_as_written():
    """v_permlane16_swap_b32 / v_swap_b32 write BOTH operands (conv_wup2 and conv_wino2 issue such swaps around their 16-byte
    stores): a swap whose SECOND operand is a store's data register in the next slot is a hazard too; behind one wait state not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_store_hazard as ch
    assert ch.written("v_permlane16_swap_b32", ["v9", "v3"]) == [(9, 9), (3, 3)]
    assert ch.written("v_add_co_u32", ["v1", "vcc", "v2", "v3"]) == [(1, 1)] and ch.written("v_cmp_lt_f32", ["vcc", "v1", "v2"]) == []
    store = (0, "buffer_store_dwordx4", ["v[0:3]", "v8", "s[4:7]", "s9", "offen"])
    swap = (8, "v_permlane16_swap_b32", ["v9", "v3"])
    end = (24, "s_endpgm", [])
    notes = []
    assert len(ch.check({"k": [store, swap, end]}, "x.o", notes)) == 1
    notes = []
    assert ch.check({"k": [store, (8, "s_nop", ["0"]), (12, "v_permlane16_swap_b32", ["v9", "v3"]), end]}, "x.o", notes) == [] and len(notes) == 1


WAR = os.path.join(ROOT, "tools", "check_mfma_war.py")


def test_no_early_write_of_an_mfma_source_operand_in_the_built_libraries(tmp_path):
    """tools/check_mfma_war.py (round 6): no VALU write of a 128-bit MFMA source operand in the window behind the MFMA, in any kernel of the
    built libraries; and the positive control -- conv_wino3.hip with the pin of position 6's dead B operand taken out again
    (IPDM_WINO3_DBG & 16384) is flagged in all four instantiations: `v_cndmask_b32_e64 v196, 0, 1, ..` three slots behind the last MFMA of
    the position's accumulate chain."""
    objs = sorted(glob.glob(os.path.join(CSRC, "*.o")))
    assert len(objs) >= 14, "csrc/*.o missing: run __graft_entry__.build()"
    r = subprocess.run([sys.executable, WAR] + objs, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 VALU write(s)" in r.stdout, r.stdout[-2000:]
    obj = str(tmp_path / "wino3_hole.o")
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-function", "-DIPDM_WINO3_DBG=16384", "-I" + CSRC, "-c",
                    os.path.join(CSRC, "conv_wino3.hip"), "-o", obj], check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, WAR, obj], capture_output=True, text=True, timeout=600)
    flagged = [ln for ln in r.stdout.splitlines() if ln.startswith("MFMA SOURCE WAR")]
    assert r.returncode == 1 and len(flagged) >= 4, r.stdout[-2000:]
    assert all("v_mfma_f32_32x32x16_bf16" in ln and "v_cndmask_b32" in ln for ln in flagged), r.stdout[-2000:]
    assert len({ln.split("conv_wino3_kernelIL")[1][:6] for ln in flagged}) == 4, flagged      # every instantiation

