#!/bin/bash
# Builds a VARIANT of libpano_hip.so beside the product one, for A/B runs on the GPU box (PANO_LIB=<path> selects it in the
# Python mirror; the product library is never touched):
#   tools/build_variant.sh <name> [<git-rev> | <patch-file>]
#     no 2nd argument: the working tree as it stands
#     <git-rev>      : csrc/ + include/ of that commit (e.g. HEAD~3: "before" of an A/B)
#     <patch-file>   : the working tree with experiments/<x>.patch applied (timing-only scratch variants live there, never in csrc/)
# -> experiments/_build/libpano_<name>.so (git-ignored; travels to the GPU box with the snapshot)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; what=$2
[ -n "$name" ] || { echo "usage: $0 <name> [<git-rev> | <patch-file>]"; exit 2; }
W=$(mktemp -d /tmp/pv_${name}_XXXX)
mkdir -p "$W/img-stitching_amd" "$ROOT/experiments/_build"
if [ -n "$what" ] && [ ! -f "$what" ]; then
  git -C "$ROOT" archive "$what" img-stitching_amd/csrc include | tar -x -C "$W"
else
  cp -r "$ROOT/img-stitching_amd/csrc" "$W/img-stitching_amd/csrc"; cp -r "$ROOT/include" "$W/include"
  [ -n "$what" ] && patch -s -p1 -d "$W" < "$what"
fi
make -s -C "$W/img-stitching_amd/csrc" OUT="$ROOT/experiments/_build/libpano_${name}.so"
rm -rf "$W"
echo "experiments/_build/libpano_${name}.so"
