#!/bin/bash
# What exactly does the SLP build of ddc_fir_i8.hip need to go wrong (NOTEBOOK R5.2)?  Variants of the file compiled WITH the
# SLP vectoriser, each with one change, run through tools/i8x_debug.py 127 1 (tuned 127 taps, loaders finish).
#   build here:      tools/ubench/slp_variants.sh build     run on the box:  gpurun -- bash tools/ubench/slp_variants.sh run
set -u
cd "$(dirname "$0")/../../libperseus-sdr_amd"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -std=c++17 -Icsrc"
if [ "${1:-}" = build ]; then
  make -s -C csrc >/dev/null || exit 1
  T=$(mktemp -d)
  cp csrc/ddc_fir_i8.hip $T/slp.hip
  # V1: a long pad behind every finishing store
  sed 's/ off nt\\n\\ts_nop 1"/ off nt\\n\\ts_nop 7\\n\\ts_nop 7"/' csrc/ddc_fir_i8.hip > $T/slp_longpad.hip
  # V2: the rotation's two results cannot be packed (an empty asm on one of them between the two statements)
  python3 - $T <<'PY'
import sys
t = sys.argv[1]
s = open(t + "/slp.hip").read()
old = "        const float x = __builtin_fmaf(-v, s, u * c), y = __builtin_fmaf(v, c, u * s);"
assert old in s
new = ("        float x = __builtin_fmaf(-v, s, u * c);\n        asm volatile(\"\" : \"+v\"(x));\n"
       "        const float y = __builtin_fmaf(v, c, u * s);")
open(t + "/slp_norotpack.hip", "w").write(s.replace(old, new))
# V3: the rotation packed as hipcc does it, but 4 idle cycles between it and everything behind it
new3 = old + "\n        asm volatile(\"s_nop 4\" ::: \"memory\");"
open(t + "/slp_nopafter.hip", "w").write(s.replace(old, new3))
# V4: the two products first, 4 idle cycles, then the two FMAs (if hipcc packs both pairs, the pause sits between the packed
# multiply and the packed FMA that consumes its result)
new4 = ("        float p0 = u * c, p1 = u * s;\n        asm volatile(\"s_nop 4\" : \"+v\"(p0), \"+v\"(p1));\n"
        "        const float x = __builtin_fmaf(-v, s, p0), y = __builtin_fmaf(v, c, p1);")
open(t + "/slp_nopbetween.hip", "w").write(s.replace(old, new4))
# V5: the same split without the pause (the asm statement alone, as the control of V4)
new5 = ("        float p0 = u * c, p1 = u * s;\n        asm volatile(\"\" : \"+v\"(p0), \"+v\"(p1));\n"
        "        const float x = __builtin_fmaf(-v, s, p0), y = __builtin_fmaf(v, c, p1);")
open(t + "/slp_splitonly.hip", "w").write(s.replace(old, new5))
PY
  for v in slp slp_longpad slp_norotpack slp_nopafter slp_nopbetween slp_splitonly; do
    /opt/rocm/bin/hipcc $FLAGS -c $T/$v.hip -o $T/$v.o 2>$T/err.txt &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_$v.so csrc/ddc_kernels.o $T/$v.o csrc/ddc_pipeline.o csrc/ddc_multi.o -L/opt/rocm/lib -lrccl &&
      echo "built ab_$v.so; packed fp32 in the layout-1 tuned-127 kernel: $(python3 - $T/$v.o <<'PY'
import sys, subprocess, re, tempfile, os
LL = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as t:
    subprocess.check_call([LL + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + t + "/f", sys.argv[1]])
    subprocess.check_call([LL + "clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + t + "/f", "--output=" + t + "/c"])
    d = subprocess.check_output([LL + "llvm-objdump", "-d", "--no-show-raw-insn", t + "/c"], text=True)
k = d.split("<_ZN4pddc9k_fir_i8xILi128ELi2ELb0ELi1ELi8EEEvNS_10FirI8xArgsExi>:")[1].split("\n\n")[0]
print(len(re.findall(r"v_pk_(mul|fma|add)_f32", k)))
PY
)" || { echo "FAILED $v"; tail -5 $T/err.txt; }
  done
  rm -rf $T
elif [ "${1:-}" = run ]; then
  cp libperseus_ddc.so /tmp/keep.so
  for v in slp slp_longpad slp_norotpack slp_nopafter slp_nopbetween slp_splitonly; do
    cp ab_$v.so libperseus_ddc.so
    echo "=== $v: batches with wrong outputs / batches run (tools/i8x_debug.py 127 1)"
    (cd .. && timeout 300 python tools/i8x_debug.py 127 1 2>&1 | grep -c "bad outputs"; )
  done
  cp /tmp/keep.so libperseus_ddc.so
fi
