#!/usr/bin/env python3
"""tools/experiments/vmm_probe.py -- feasibility of the HIP virtual-memory API on this stack (hipMemCreate / hipMemAddressReserve /
hipMemMap / hipMemSetAccess through ctypes): physical 2 GiB handles mapped whole, then pieces of two handles interleaved
into one address range; the traffic-only probe runs on both."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.placement import CHUNK_BYTES, plane_slots  # noqa: E402


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Flags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32", C.c_void_p), ("allocFlags", Flags)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def main():
    hip = C.CDLL("libamdhip64.so")
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    prop = Prop(type=1, requestedHandleType=0, location=Loc(1, 0), win32=None)
    gran = C.c_size_t(0)
    print("granularity rc", hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 0), gran.value)
    rec = C.c_size_t(0)
    print("recommended rc", hip.hipMemGetAllocationGranularity(C.byref(rec), C.byref(prop), 1), rec.value)
    hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(Prop), C.c_ulonglong]
    hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Access), C.c_size_t]
    hip.hipMemUnmap.argtypes = [C.c_void_p, C.c_size_t]
    acc = Access(Loc(1, 0), 3)
    nh = 6
    handles, vas = [], []
    for i in range(nh):
        h = C.c_void_p()
        rc = hip.hipMemCreate(C.byref(h), CHUNK_BYTES, C.byref(prop), 0)
        va = C.c_void_p()
        rc2 = hip.hipMemAddressReserve(C.byref(va), CHUNK_BYTES, 0, None, 0)
        rc3 = hip.hipMemMap(va, CHUNK_BYTES, 0, h, 0)
        rc4 = hip.hipMemSetAccess(va, CHUNK_BYTES, C.byref(acc), 1)
        print("handle %d: create %d reserve %d map %d access %d  va 0x%x" % (i, rc, rc2, rc3, rc4, va.value or 0))
        handles.append(h)
        vas.append(va.value)
    w, h_, B, profile = 3840, 2160, 20, 2
    n3 = 3 * w * h_
    _, hs, st, _ = L.plane_geometry(w, h_, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    _, _, offs = plane_slots(CHUNK_BYTES, [B * x for x in psz])
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(nh):
        ctx.synth_frames_device(vas[i], n3, B, w, h_, 1, 0)
    torch.cuda.synchronize()

    def probe(src, plv):
        return ctx.probe_encode_traffic(src, n3, B, w, h_, [plv + o for o in offs], st, psz, iters=3)

    probe(vas[0], vas[1])
    for i in range(nh):
        print("input handle %d:" % i, " ".join("%.3f" % probe(vas[i], vas[j]) if i != j else "  .  " for j in range(nh)))
    # interleave handles 0 and 1 in 32 MiB stripes into one 2 GiB range (first halves of each)
    stripe = 32 << 20
    va = C.c_void_p()
    print("reserve", hip.hipMemAddressReserve(C.byref(va), CHUNK_BYTES, 0, None, 0))
    hip.hipMemUnmap(vas[0], CHUNK_BYTES)
    hip.hipMemUnmap(vas[1], CHUNK_BYTES)
    rcs = set()
    for k in range(CHUNK_BYTES // stripe):
        hsel = handles[k % 2]
        rcs.add(hip.hipMemMap(va.value + k * stripe, stripe, (k // 2) * stripe, hsel, 0))
    rcs.add(hip.hipMemSetAccess(va, CHUNK_BYTES, C.byref(acc), 1))
    print("striped map rcs", rcs)
    if rcs != {0}:
        print("hipMemMap with a non-zero offset is not supported here: no striping by sub-range mapping")
        return
    ctx.synth_frames_device(va.value, n3, B, w, h_, 1, 0)
    torch.cuda.synchronize()
    print("striped (handles 0/1) as input:", " ".join("%.3f" % probe(va.value, vas[j]) for j in range(2, nh)))


if __name__ == "__main__":
    main()
