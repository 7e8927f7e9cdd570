#!/usr/bin/env python3
"""Instruction-level bisect of the packed-fp32 fault beside matrix waves (NOTEBOOK R5.2, R6.4): ddc_fir_i8.hip compiled WITH
the SLP vectoriser delivers wrong x in lanes 48..63 of finishing loader waves (layout 1).  This builds libraries that differ
from that failing build by ONE edit of its machine code -- the device assembly is edited and taken through the rest of
hipcc's own pipeline (cc1as, lld, clang-offload-bundler, the host object's .hip_fatbin section replaced):
    slp       the failing build as hipcc makes it
    war_nop   `s_nop 3` between every  v_pk_mul_f32 .. op_sel:[0,1]  and the  v_mov_b32  behind it, which OVERWRITES the
              multiply's first source register (write after read)
    raw_nop   `s_nop 3` in front of the dependent  v_pk_fma_f32 .. neg_lo  instead (read after write)
    no_mov    the v_mov_b32 removed: the packed FMA takes the high register for both halves itself (op_sel:[1,0,0]
              op_sel_hi:[1,1,1]) -- no write to the multiply's source at all, the same arithmetic
    mov_late  the v_mov_b32 moved down to just in front of the packed FMA (one instruction further from the multiply)
    fma_scalar / mul_scalar   one of the two packed instructions as two scalar ones, the other left packed
    mul_nosel  the multiply packed but without the half swap of its second source (built with two v_mov first)
    nop_before `s_nop 3` in front of the multiply (behind the vector instructions that make its operands)
    two_movs_before / sel_on_copy / sel_src0   separate mul_nosel's three differences: two more vector instructions in front,
              the copy of the second source, the half swap by op_sel
    pad_vgpr_<n>  no instruction changed: the clean two-k-step kernel (48 taps) given n VGPRs (the failing kernel has 155); run with
              `python tools/i8x_debug.py 48 1`
    as_fma    the multiply as v_pk_fma_f32 .., 0 with the same op_sel on src1
    vmcnt0_before  `s_waitcnt vmcnt(0)` in front of the multiply: the wave's own global loads (the next tile's, issued just
              before the finishing code) have all returned -- is a load's register write part of it?
    mfma_zero_reg  the packed code untouched; the MATRIX waves' accumulation-starting instructions read their zero from four
              registers instead of the inline constant 0 (the wrong product is exactly 0.0: does the neighbour's constant leak?)
usage: python tools/ubench/pk_asm_variants.py build       (here, no GPU)   -> libperseus-sdr_amd/ab_<variant>.so
       gpurun -- bash tools/ab.sh run 1 -- python tools/i8x_debug.py 127 1 (on the box; `done` without `bad outputs` = clean)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "libperseus-sdr_amd")
CSRC = os.path.join(PKG, "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
DEV = "ddc_fir_i8-hip-amdgcn-amd-amdhsa-gfx950"

MUL = re.compile(r"^\tv_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\] op_sel_hi:\[0,0\]\s*$")
MOV = re.compile(r"^\tv_mov_b32_e32 v(\d+), v(\d+)\s*$")
FMA = re.compile(r"^\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel_hi:\[0,1,1\] neg_lo:\[1,0,0\]\s*$")


def edit(lines, variant):
    """returns the edited lines and the number of sites changed"""
    out, n, i = [], 0, 0
    while i < len(lines):
        m = MUL.match(lines[i])
        if not m or variant == "slp":
            out.append(lines[i])
            i += 1
            continue
        # the site: mul, [mov of src0.hi into src0.lo], ..., fma (within a few instructions)
        j_mov = j_fma = None
        for j in range(i + 1, min(i + 8, len(lines))):
            mm = MOV.match(lines[j])
            if mm and j_mov is None and mm.group(1) == m.group(3) and mm.group(2) == m.group(4):
                j_mov = j
            if FMA.match(lines[j]):
                j_fma = j
                break
        if j_mov is None or j_fma is None:
            out.append(lines[i])
            i += 1
            continue
        n += 1
        seg = lines[i:j_fma + 1]
        k_mov, k_fma = j_mov - i, j_fma - i
        if variant == "war_nop":
            seg = seg[:1] + ["\ts_nop 3"] + seg[1:]
        elif variant == "raw_nop":
            seg = seg[:k_fma] + ["\ts_nop 3"] + seg[k_fma:]
        elif variant == "no_mov":
            f = FMA.match(seg[k_fma])
            seg[k_fma] = (f"\tv_pk_fma_f32 v[{f.group(1)}:{f.group(2)}], v[{f.group(3)}:{f.group(4)}], v[{f.group(5)}:{f.group(6)}], "
                          f"v[{f.group(7)}:{f.group(8)}] op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0]")
            del seg[k_mov]
        elif variant == "mov_late":
            mv = seg.pop(k_mov)
            seg.insert(k_fma - 1, mv)
        elif variant == "fma_scalar":           # the packed multiply stays; the FMA as two v_fma_f32 (and no v_mov)
            f = FMA.match(seg[k_fma])
            d0, d1, a1 = f.group(1), f.group(2), m.group(4)        # a1: the register that holds v (src0's high half)
            seg[k_fma:k_fma + 1] = [f"\tv_fma_f32 v{d0}, -v{a1}, v{f.group(5)}, v{f.group(7)}",
                                    f"\tv_fma_f32 v{d1}, v{a1}, v{f.group(6)}, v{f.group(8)}"]
            del seg[k_mov]
        elif variant == "mul_scalar":           # the packed FMA stays; the multiply as two v_mul_f32
            seg[0:1] = [f"\tv_mul_f32_e32 v{m.group(1)}, v{m.group(3)}, v{m.group(6)}",
                        f"\tv_mul_f32_e32 v{m.group(2)}, v{m.group(3)}, v{m.group(5)}"]
        elif variant == "mul_nosel":            # the multiply packed, but without swapping its second source's halves: the swapped
            # pair is built in the destination registers first (two v_mov), op_sel only broadcasts src0's low half
            seg[0:1] = [f"\tv_mov_b32_e32 v{m.group(1)}, v{m.group(6)}", f"\tv_mov_b32_e32 v{m.group(2)}, v{m.group(5)}",
                        f"\tv_pk_mul_f32 v[{m.group(1)}:{m.group(2)}], v[{m.group(3)}:{m.group(4)}], v[{m.group(1)}:{m.group(2)}] op_sel_hi:[0,1]"]
        elif variant == "two_movs_before":      # the failing multiply untouched, behind the two v_mov that mul_nosel needs (timing only)
            seg = [f"\tv_mov_b32_e32 v{m.group(1)}, v{m.group(6)}", f"\tv_mov_b32_e32 v{m.group(2)}, v{m.group(5)}"] + seg
        elif variant == "sel_on_copy":          # mul_nosel's timing and registers, but the half swap done by op_sel again
            seg[0:1] = [f"\tv_mov_b32_e32 v{m.group(1)}, v{m.group(5)}", f"\tv_mov_b32_e32 v{m.group(2)}, v{m.group(6)}",
                        f"\tv_pk_mul_f32 v[{m.group(1)}:{m.group(2)}], v[{m.group(3)}:{m.group(4)}], v[{m.group(1)}:{m.group(2)}] op_sel:[0,1] op_sel_hi:[0,0]"]
        elif variant == "sel_src0":             # the same product with the sources exchanged: the swap sits on src0
            seg[0:1] = [f"\tv_pk_mul_f32 v[{m.group(1)}:{m.group(2)}], v[{m.group(5)}:{m.group(6)}], v[{m.group(3)}:{m.group(4)}] op_sel:[1,0] op_sel_hi:[0,0]"]
        elif variant == "as_fma":               # the same product as a packed FMA with a zero addend: is it the multiply, or VOP3P's src1?
            seg[0:1] = [f"\tv_pk_fma_f32 v[{m.group(1)}:{m.group(2)}], v[{m.group(3)}:{m.group(4)}], v[{m.group(5)}:{m.group(6)}], 0 "
                        f"op_sel:[0,1,0] op_sel_hi:[0,0,0]"]
        elif variant == "vmcnt0_before":        # no global load of this wave in flight while the multiply reads its operands
            seg = ["\ts_waitcnt vmcnt(0)"] + seg
        elif variant == "nop15_before":         # sixteen idle cycles: a quarter-rate v_mul_lo_u32 three instructions up has drained
            seg = ["\ts_nop 15"] + seg
        elif variant == "nop_before":           # idle cycles between the VALU instructions that make c and s and the multiply
            seg = ["\ts_nop 3"] + seg
        out += seg
        i = j_fma + 1
    return out, n


KNAME = "_ZN4pddc9k_fir_i8xILi128ELi2ELb0ELi1ELi8EEEvNS_10FirI8xArgsExi"      # tuned, <= 128 taps, layout 1: what i8x_debug.py 127 1 runs


def edit_mfma_zero(lines):
    """variant mfma_zero_reg: in the failing kernel the matrix instructions that start an accumulation take their zero from
    four REGISTERS (v[156:159], written once at the kernel's entry; the kernel used 155) instead of the inline constant 0"""
    out, n, inside = [], 0, False
    for l in lines:
        if l.startswith(KNAME + ":"):
            inside = True
            out.append(l)
            out += [f"\tv_mov_b32_e32 v{r}, 0" for r in range(156, 160)]
            continue
        if inside and ".end_amdhsa_kernel" in l:
            inside = False
        if inside:
            if re.match(r"^\tv_mfma_i32_16x16x64_i8 .*, 0\s*$", l):
                l = re.sub(r", 0\s*$", ", v[156:159]", l)
                n += 1
            elif ".amdhsa_next_free_vgpr 155" in l:
                l = l.replace("155", "160")
            elif ".amdhsa_accum_offset 156" in l:
                l = l.replace("156", "160")
        out.append(l)
    return out, n


def edit_pad_vgpr(lines, nfree):
    """variant pad_vgpr_<n>: NO instruction changed -- the two-k-step tuned kernel (<= 64 taps, layout 1: the same finishing code,
    the same registers in the packed multiply, 113 VGPRs, clean with the SLP build) is given the register ALLOCATION of the
    failing three-k-step kernel (155): does the fault follow the allocation (where a wave's registers lie in the file)?"""
    k64 = KNAME.replace("ILi128E", "ILi64E")
    out, inside, n = [], False, 0
    for l in lines:
        if l.startswith(k64 + ":"):
            inside = True
        elif inside and ".end_amdhsa_kernel" in l:
            inside = False
        if inside and ".amdhsa_next_free_vgpr" in l:
            l = re.sub(r"\d+", str(nfree), l)
            n += 1
        elif inside and ".amdhsa_accum_offset" in l:
            l = re.sub(r"\d+", str(nfree // 4 * 4), l)
        out.append(l)
    return out, n


def sh(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(f"FAILED: {' '.join(cmd)}\n{r.stderr[-2000:]}")


def main():
    if len(sys.argv) < 2 or sys.argv[1] != "build":
        sys.exit(__doc__)
    subprocess.check_call(["make", "-s", "-C", CSRC])
    t = tempfile.mkdtemp()
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".hip", ".inc")):
            shutil.copy(os.path.join(CSRC, f), t)
    # the failing build: the Makefile's flags WITHOUT -fno-slp-vectorize; every intermediate file kept
    sh(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-fvisibility=hidden", "-std=c++17", "-c", "ddc_fir_i8.hip",
        "-o", "slp.o", "-save-temps"], t)
    src = open(os.path.join(t, DEV + ".s")).read().split("\n")
    for f in os.listdir(PKG):
        if f.startswith("ab_") and f.endswith(".so"):
            os.remove(os.path.join(PKG, f))
    for v in (sys.argv[2:] or ("slp", "war_nop", "raw_nop", "no_mov", "mov_late", "fma_scalar", "mul_scalar", "mul_nosel", "nop_before")):
        if v.startswith("pad_vgpr_"):
            lines, n = edit_pad_vgpr(list(src), int(v.split("_")[-1]))
        else:
            lines, n = edit_mfma_zero(list(src)) if v == "mfma_zero_reg" else edit(list(src), v)
        open(os.path.join(t, v + ".s"), "w").write("\n".join(lines))
        sh([LLVM + "/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-target-cpu", "gfx950", "-mrelocation-model",
            "pic", "-o", v + ".dev.o", v + ".s"], t)
        sh([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-plugin-opt=-amdgpu-internalize-symbols",
            "-plugin-opt=mcpu=gfx950", "-plugin-opt=O3", "--whole-archive", "-o", v + ".out", v + ".dev.o", "--no-whole-archive"], t)
        sh([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096",
            "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + v + ".out",
            "-output=" + v + ".hipfb"], t)
        sh([LLVM + "/llvm-objcopy", "--update-section", ".hip_fatbin=" + v + ".hipfb", "slp.o", v + ".o"], t)
        sh(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(PKG, f"ab_{v}.so"),
            os.path.join(CSRC, "ddc_kernels.o"), v + ".o", os.path.join(CSRC, "ddc_pipeline.o"), os.path.join(CSRC, "ddc_multi.o"),
            "-L/opt/rocm/lib", "-lrccl"], t)
        print(f"built ab_{v}.so ({n} sites edited)")
    shutil.rmtree(t)


if __name__ == "__main__":
    main()
