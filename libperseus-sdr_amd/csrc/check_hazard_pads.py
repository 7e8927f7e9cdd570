#!/usr/bin/env python3
"""Build-time guard for ddc_fir_i8.o (run by the Makefile right after the compile; a failure deletes the object).

k_fir_i8x's loader / finishing waves share their SIMD with waves that issue matrix instructions.  In that position two
code shapes gave wrong values in lanes 48..63 on MI355X (NOTEBOOK.md R4.4, R5.2: the product compiled WITH the SLP
vectoriser delivers 5-6 thousand wrong outputs per 600-tile batch in layout 1, x only, lanes 48..63; the isolated
instruction sequences are clean -- tools/ubench/mfma_*_hazard.hip -- so the cause is not understood, only avoided):
  (b) packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) in a wave beside a matrix wave,
  (a) a result store whose data registers are overwritten at once.
The source avoids both (no 2-vector arithmetic in the finishing code, -fno-slp-vectorize for the file, store + s_nop in ONE
asm statement).  A compiler update, a build line without the flag or a new 2-vector expression would bring them back
silently; this script looks at the machine code instead:
  * every k_fir_i8x / k_fir_i8x_many instantiation with LAYOUT 1 (loaders finish tiles) or LAYOUT 2 (two finishing waves
    whose SIMDs hold no matrix wave only as long as waves w, w + 4, w + 8 share a SIMD -- measured, not guaranteed): no
    v_pk_*_f32 at all;
  * LAYOUT 1 and 2: every nontemporal result store (`global_store_dword[x2] ... nt`) is followed by `s_nop`, no global
    store at all is followed directly by a vector instruction, and there IS at least one such store (a kernel without
    any would pass the check by having lost what it checks).
usage: check_hazard_pads.py ddc_fir_i8.o [--arch gfx950] [--llvm-bin DIR]"""
import os
import re
import subprocess
import sys
import tempfile

NAME = re.compile(r"k_fir_i8x(?:_many)?ILi(\d+)ELi(\d)ELb([01])ELi(\d)E")


class ToolError(Exception):
    pass


def disassemble(obj, arch, llvm):
    tools = {t: os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")}
    for t, path in tools.items():
        if not os.path.exists(path):
            raise ToolError(f"{t} not found in {llvm} (pass --llvm-bin, or set ROCM_LLVM_BIN)")
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "dev.co")
        r = subprocess.run([tools["llvm-objcopy"], "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True, text=True)
        if r.returncode or not os.path.exists(fat):
            raise ToolError(f"{obj} holds no .hip_fatbin section: {r.stderr.strip()}")
        target = f"hipv4-amdgcn-amd-amdhsa--{arch}"
        r = subprocess.run([tools["clang-offload-bundler"], "--type=o", "--unbundle", f"--targets={target}", f"--input={fat}",
                            f"--output={co}"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            raise ToolError(f"no code object for {target} in {obj} (built for another --offload-arch?): {r.stderr.strip()}")
        return subprocess.check_output([tools["llvm-objdump"], "-d", "--no-show-raw-insn", co], text=True)


def main(argv):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("--arch", default="gfx950")
    ap.add_argument("--llvm-bin", default=os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin"))
    ap.add_argument("--report-src1-swap", action="store_true",
                    help="no check, a count: packed fp32 instructions whose SECOND source takes its low half from the high register\n"
                         "(op_sel[1] = 1) -- the operand selection that delivered a zero low half beside a matrix wave (NOTEBOOK R6.4);\n"
                         "for objects whose kernels never share a SIMD with a matrix wave inside one launch chain (ddc_kernels.o)")
    a = ap.parse_args(argv)
    if a.report_src1_swap:
        try:
            text = disassemble(a.obj, a.arch, a.llvm_bin)
        except ToolError as e:
            print("check_hazard_pads: cannot report:", e, file=sys.stderr)
            return 0
        pk = [l for l in text.splitlines() if re.search(r"\bv_pk_(mul|fma|add)_f32\b", l)]
        sw = [l for l in pk if re.search(r"op_sel:\[[01],1", l)]
        print(f"check_hazard_pads: {os.path.basename(a.obj)}: {len(pk)} packed fp32 instructions, {len(sw)} with src1's low half from the "
              f"high register (harmless unless a wave on the same SIMD issues matrix instructions: NOTEBOOK R6.4)")
        return 0
    try:
        text = disassemble(a.obj, a.arch, a.llvm_bin)
    except ToolError as e:
        print("check_hazard_pads: cannot check:", e, file=sys.stderr)
        return 2
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
        elif cur is not None and line.startswith("\t"):
            ins = line.strip().split("//")[0].strip()
            if ins:
                cur.append(ins)
    problems, checked = [], 0
    for name, code in kernels.items():
        m = NAME.search(name)
        if not m:
            continue
        layout = int(m.group(4))
        if layout == 0:
            continue                       # the matrix waves finish their own tiles: no other matrix wave on their SIMD
        checked += 1
        pk = [i for i in code if re.match(r"v_pk_(mul|fma|add)_f32", i)]
        if pk:
            problems.append(f"{name}: {len(pk)} packed fp32 instructions in a kernel whose finishing waves may sit beside matrix "
                            f"waves (first: {pk[0]})")
        nt = 0
        for k, ins in enumerate(code[:-1]):
            if not ins.startswith("global_store"):
                continue
            nxt = code[k + 1]
            if re.match(r"global_store_dword(x2)? .* nt$", ins):
                nt += 1
                if not nxt.startswith("s_nop"):
                    problems.append(f"{name}: result store without its pad: `{ins}` then `{nxt}`")
            elif nxt.startswith("v_"):
                problems.append(f"{name}: a vector instruction directly behind a store: `{ins}` then `{nxt}`")
        if not nt:
            problems.append(f"{name}: no nontemporal result store found at all: has store_f2_padded changed its instruction?")
    if not checked:
        problems.append("no k_fir_i8x instantiation with LAYOUT 1 or 2 found: has the kernel been renamed?")
    for p in problems:
        print("check_hazard_pads:", p, file=sys.stderr)
    if not problems:
        print(f"check_hazard_pads: {checked} k_fir_i8x kernels with finishing waves (layouts 1, 2): no packed fp32, every result store padded")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
