#!/bin/bash
# Register / LDS / spill figures of every kernel in a HIP object (from the code object's metadata notes).
# usage: tools/kernel_resources.sh libperseus-sdr_amd/csrc/ddc_kernels.o [name-substring]
set -e
LLVM=${ROCM_LLVM_BIN:-/opt/rocm/lib/llvm/bin}
T=$(mktemp -d); trap 'rm -rf $T' EXIT
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
$LLVM/clang-offload-bundler --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co
$LLVM/llvm-readelf --notes $T/dev.co | python3 -c '
import sys, re
sel = sys.argv[1] if len(sys.argv) > 1 else ""
cur = {}
def flush():
    if cur.get(".name") and sel in cur[".name"]:
        print("%-90s vgpr %3s agpr %3s sgpr %3s spill v %3s s %3s lds %6s scratch %s" % (cur[".name"][:90], cur.get(".vgpr_count"), cur.get(".agpr_count"), cur.get(".sgpr_count"), cur.get(".vgpr_spill_count"), cur.get(".sgpr_spill_count"), cur.get(".group_segment_fixed_size"), cur.get(".private_segment_fixed_size")))
for line in sys.stdin:
    m = re.match(r"\s+(- )?(\.[a-z_]+):\s+(.*)", line)
    if not m: continue
    k, v = m.group(2), m.group(3).strip().strip("\x27")
    if k == ".agpr_count" and cur.get(".agpr_count") is not None:
        flush(); cur.clear()
    if k in (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size", ".private_segment_fixed_size"):
        if k == ".name" and v.startswith("_Z") is False and cur.get(".name"): continue
        cur[k] = v
flush()
' "$2"
