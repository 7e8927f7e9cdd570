#!/bin/bash
# Same-box A/B of builds of the kernel library (boxes differ by a few %, so two versions are only ever compared on ONE box,
# alternating).  The alternatives are built NEXT TO the product (libperseus-sdr_amd/ab_<name>.so) and selected through
# PDDC_DDC_LIB, which the ctypes binding honours -- the product library is never overwritten.
#   build here (no GPU):   tools/ab.sh build <spec> ...
#        name            the working tree
#        name@REV        the tree of git revision REV (HEAD, HEAD~2, a hash)
#        name:FLAGS      the working tree with extra compiler flags (e.g. nopad:"-DX=1"); name@REV:FLAGS works too
#   run on the GPU box:    gpurun -- bash tools/ab.sh run [REPS] -- <command ...>      (e.g. python tools/state_2p28.py "pair api k_fir8")
#        every line the command prints comes back prefixed with the build's name
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/libperseus-sdr_amd"
if [ "${1:-}" = build ]; then
  shift
  for spec in "$@"; do
    flags=""; case "$spec" in *:*) flags=${spec#*:}; spec=${spec%%:*};; esac
    rev=""; case "$spec" in *@*) rev=${spec#*@}; spec=${spec%%@*};; esac
    name=$spec
    T=$(mktemp -d)
    mkdir -p "$T/libperseus-sdr_amd"
    if [ -n "$rev" ]; then
      (cd "$ROOT" && git archive "$rev" libperseus-sdr_amd/csrc include) | tar -x -C "$T" || { echo "FAILED $name: no such revision $rev"; continue; }
    else
      cp -r "$PKG/csrc" "$T/libperseus-sdr_amd/csrc" && cp -r "$ROOT/include" "$T/include" && rm -f "$T"/libperseus-sdr_amd/csrc/*.o
    fi
    if make -s -C "$T/libperseus-sdr_amd/csrc" EXTRA="$flags" ../libperseus_ddc.so >"$T/log.txt" 2>&1; then
      cp "$T/libperseus-sdr_amd/libperseus_ddc.so" "$PKG/ab_$name.so" && echo "built ab_$name.so (${rev:-working tree}${flags:+, $flags})"
    else
      echo "FAILED $name"; tail -15 "$T/log.txt"
    fi
    rm -rf "$T"
  done
elif [ "${1:-}" = run ]; then
  shift
  reps=2
  if [ "${1:-}" != "--" ]; then reps=$1; shift; fi
  [ "${1:-}" = "--" ] && shift
  cd "$ROOT"
  for rep in $(seq 1 "$reps"); do
    for f in "$PKG"/ab_*.so; do
      n=$(basename "$f" .so); n=${n#ab_}
      PDDC_DDC_LIB="$f" timeout "${AB_TIMEOUT:-600}" "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/[$n] /"
    done
  done
else
  echo "usage: $0 build <name[@rev][:flags]> ... | run [reps] -- <command>"; exit 2
fi
