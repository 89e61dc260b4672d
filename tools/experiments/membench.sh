#!/bin/bash
# tools/experiments/membench.sh <variant 1..7> [program arguments] -- ONE entry point for the seven memory-system probes
# (run on the GPU box): builds membench<variant>.hip for gfx950 into $TMPDIR and runs it.  What each variant varies:
#   1 float4 copy, encode-shaped and decode-shaped traffic (the chip's streaming ceiling for each mix)
#   2 encode mix: waves side by side per workgroup (256x8 / 512x4 / 1024x2 px tiles), non-temporal accesses, store widths
#   3 encode mix: units in flight per thread, XCD-aware tile order, workgroup size
#   4 decode mix (3 B read + 12 B written): plain against non-temporal accesses, 4 against 8 pixels per thread and row
#   5 encode mix: loads only / stores only / both, with the kernel's own tile walk
#   6 encode mix: general workgroup tile (WX waves in x, WY in y)
#   7 encode mix: cache-policy bits (sc0 / sc1 / nt) on loads and stores
# Results: profiles/r01_membench.txt (1-4), profiles/r02_membench.txt (5-7).
set -eu
V=${1:?usage: membench.sh <1..7> [args]}; shift || true
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/membench$([ "$V" = 1 ] && echo "" || echo "$V").hip
[ -f "$SRC" ] || { echo "no such variant: $V" >&2; exit 2; }
OUT=${TMPDIR:-/tmp}/membench$V
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o "$OUT" "$SRC"
exec "$OUT" "$@"
