#!/usr/bin/env python3
"""Capture the PSI (Ferwerda t.v.i.) and JND HDR-VDP luminance tables as binary data.

The reference compiles these six tables in as data (include/luma/ptfs/*.h, `#include`d at
include/luma/luma_quantizer.h:55-77); they are the definition of two of its five transfer
functions and cannot be regenerated from a formula in the tree ("PTF tables are
data").  This script reads them back *through the reference's own API* -- LumaQuantizer::getMapping()
of oracle/_ref/libluma_ref.so after setQuantizer(PTF_PSI|PTF_JND_HDRVDP, 10|11|12, ...) -- and stores the
float32 values little-endian, nothing else, as lumahdrv_amd/data/ptf_<name>_<bits>.f32.

Runs only in the build container (needs /root/reference for `make -C oracle ref`).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as o  # noqa: E402


def main():
    o.build(ref=True)
    out = os.path.join(ROOT, "lumahdrv_amd", "data")
    os.makedirs(out, exist_ok=True)
    for name, ptf in (("psi", o.PTF_PSI), ("jnd_hdrvdp", o.PTF_JND_HDRVDP)):
        for bits in (10, 11, 12):
            m = o.RefQuantizer(ptf, bits, o.CS_LUV, 8, 1e4, 0.005).mapping
            assert m.size == 1 << bits
            path = os.path.join(out, "ptf_%s_%d.f32" % (name, bits))
            m.astype("<f4").tofile(path)
            print(path, m.size, "floats", m[0], "...", m[-1], "fnv", "%016x" % o.fnv1a64(m))


if __name__ == "__main__":
    main()
