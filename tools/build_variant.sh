#!/bin/bash
# Cross-compiles the CURRENT tree's library into tools/_ab/libpgtwin_<name>.so for tools/lib_ab.py (A/B of two builds
# in one process).  Only the translation units that changed are recompiled (objects cached in tools/_ab/obj, keyed by
# a content hash of the source + headers + extra flags).   usage: bash tools/build_variant.sh <name> [extra hipcc flags]
set -eu
NAME=$1; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=${PGT_CSRC:-$ROOT/popgenomicstools_amd/csrc}   # PGT_CSRC: a patched copy of csrc/ (one-off experiments: see profiles/r06/af8_issue_stall.md)
OBJ=$ROOT/tools/_ab/obj
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -I$CSRC $*"
objs=()
for src in pgt_kernels.hip pgt_af_kernels.hip pgt_ingest.hip pgt_api.cpp pgt_windows.cpp; do
  key=$( (echo "$FLAGS"; cat "$CSRC/$src" "$CSRC"/*.h "$ROOT/include/pgtwin.h") | sha256sum | cut -c1-16)
  o=$OBJ/${src%.*}.$key.o
  if [ ! -f "$o" ]; then hipcc $FLAGS -x hip -c "$CSRC/$src" -o "$o" & fi
  objs+=("$o")
done
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o "$ROOT/tools/_ab/libpgtwin_$NAME.so" "${objs[@]}"
echo "built tools/_ab/libpgtwin_$NAME.so"
