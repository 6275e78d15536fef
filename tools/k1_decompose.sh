#!/bin/bash
# What K1 waits for (docs/EXPERIMENTS.md, round 5): the shipped warp kernel against two TIMING-ONLY builds of it -
#   loads only  : scalar prologue, table entry, box copy into LDS, barrier; no taps, no arithmetic, no stores
#   no box copy : scalar prologue, table entry, barrier, taps from whatever LDS holds, arithmetic, stores
# (wrong pictures on purpose).  Build here (hipcc, no GPU needed):  bash tools/k1_decompose.sh build
# then on the GPU box:  gpurun -- 'bash tools/k1_decompose.sh'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  mkdir -p "$ROOT/experiments/_build"
  for v in loads_only no_box_copy; do
    W=$(mktemp -d /tmp/k1d_${v}_XXXX); mkdir -p "$W/img-stitching_amd"; cp -r "$ROOT/img-stitching_amd/csrc" "$W/img-stitching_amd/csrc"; cp -r "$ROOT/include" "$W/include"
    python3 - "$W" "$v" <<'PY'
import sys
p = sys.argv[1] + '/img-stitching_amd/csrc/pano_warp.hip'
s = open(p).read()
i = s.index("void warp_tiles_lut_kernel(")
j = s.index("void launch_warp_tiles(")
k = s[i:j]
if sys.argv[2] == "loads_only":
    a = "    if (bh || (GAIN && gbase >= 0)) __syncthreads();  // workgroup-uniform\n    if (!active) return;"
    assert a in k
    k = k.replace(a, "    if (bh || (GAIN && gbase >= 0)) __syncthreads();  // workgroup-uniform\n    if (e.x != 0x12345678u || !active) return;")
else:
    a = "    if (bh) {\n        // chunk k = tid + 256 * it -> (row r = k / cpr, column ci = k % cpr), copied by global_load_lds_dwordx4"
    assert a in k
    k = k.replace(a, "    if (bh && stride == 1u) {\n        // chunk k = tid + 256 * it")
open(p, 'w').write(s[:i] + k + s[j:])
PY
    make -s -C "$W/img-stitching_amd/csrc" OUT="$ROOT/experiments/_build/libpano_k1_${v}.so"
    rm -rf "$W"; echo "experiments/_build/libpano_k1_${v}.so"
  done
  exit 0
fi
cd "$ROOT"
echo "product:";                                         python tools/frames_one_at_a_time.py
echo "K1 loads only (timing only):";  PANO_LIB=experiments/_build/libpano_k1_loads_only.so  python tools/frames_one_at_a_time.py
echo "K1 no box copy (timing only):"; PANO_LIB=experiments/_build/libpano_k1_no_box_copy.so python tools/frames_one_at_a_time.py
