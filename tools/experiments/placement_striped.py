#!/usr/bin/env python3
"""tools/experiments/placement_striped.py -- experiment (needs lumahdrv_amd/lib_exp/liblumahip.so: apply tools/experiments/placement_striped.patch, then
`make OUT=$PWD/../lib_exp EXTRA=-DLH_EXP_CHAN_STRIDE $PWD/../lib_exp/liblumahip.so` in lumahdrv_amd/csrc, then revert the patch):
what would the kernels gain if the three colour planes of the float frames lived in three different region groups?
The channel stride of the kernels is overridden so that R, G and B planes sit in three 2 GiB chunks."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LUMAHIP_LIB"] = os.path.join(ROOT, "lumahdrv_amd", "lib_exp", "liblumahip.so")
os.environ["LUMAHIP_TUNING"] = "1"   # the LUMAHIP_* overrides are honoured only under this gate
os.environ["LUMAHIP_ALLOW_ALIASED_FRAMES"] = "1"      # frame stride = one plane: the layout check would refuse it
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd import capi  # noqa: E402
from lumahdrv_amd.placement import CHUNK_BYTES, find_groups, plane_slots  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    w, h, B, profile = 3840, 2160, 20, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    _, _, offs = plane_slots(CHUNK_BYTES, [B * x for x in psz])
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = capi.lib()
    lib.lumahip_exp_set_chan_stride.argtypes = [ctypes.c_ulonglong]
    chunks = [torch.zeros(CHUNK_BYTES, dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize()
    va = [c.data_ptr() for c in chunks]
    print("VA steps (GiB):", sorted(set((va[i] - va[i + 1]) / 2**30 for i in range(n - 1))))

    def t4(i, y, u, v, enc=False, fs=n3, direction=0):
        pl = [chunks[y].data_ptr() + offs[0], chunks[u].data_ptr() + offs[1], chunks[v].data_ptr() + offs[2]]
        if enc:
            return ctx.time_launches(direction, 3, va[i] if isinstance(i, int) else i, fs, B, w, h, 1.0, profile, pl, st, psz)
        return ctx.probe_encode_traffic(va[i] if isinstance(i, int) else i, fs, B, w, h, pl, st, psz, iters=3)

    t4(1, 0, 0, 0)
    groups, fast, _ = find_groups(n, lambda i, r: t4(i, r, r, r))
    gid = {}
    for g, grp in enumerate(groups):
        for i in grp:
            gid[i] = g
    print("group sizes", [len(g) for g in groups])
    # three chunks equally spaced in VA (descending with the index), pairwise different groups, plus plane chunks
    best = None
    for k in range(1, n // 2):
        for i in range(n - 2 * k):
            a, b, c = i + 2 * k, i + k, i      # ascending VA: a < b < c
            if va[b] - va[a] == va[c] - va[b] and len({gid[a], gid[b], gid[c]}) == 3:
                best = (a, b, c)
                break
        if best:
            break
    if not best:
        print("no equally spaced triple of three groups")
        return
    a, b, c = best
    stride = (va[b] - va[a]) // 4
    ga, gb, gc = gid[a], gid[b], gid[c]
    others = [i for i in range(n) if i not in best]
    pick = lambda g: [i for i in others if gid[i] == g]
    print("R, G, B planes in chunks %d, %d, %d (groups %d %d %d), channel stride %.0f GiB" % (a, b, c, ga, gb, gc, stride * 4 / 2**30))
    ya, yb, yc = pick(ga)[0], pick(gb)[0], pick(gc)[0]
    ua, ub, uc = pick(ga)[1], pick(gb)[1], pick(gc)[1]
    rows = []
    lib.lumahip_exp_set_chan_stride(0)
    rows.append(("packed frames in A; Y in B, U V in C", t4(a, yb, uc, uc), t4(a, yb, uc, uc, True)))
    lib.lumahip_exp_set_chan_stride(stride)
    fs = w * h           # frame f's R plane at base + f*w*h; G and B one channel stride further each
    for name, (y, u, v) in (("R G B in A B C; Y U V together in A", (ya, ya, ya)), ("R G B in A B C; Y in A, U V in B", (ya, ub, ub)),
                            ("R G B in A B C; Y in A, U in B, V in C", (ya, ub, uc)), ("R G B in A B C; Y in B, U V in C", (yb, uc, uc))):
        rows.append((name, t4(a, y, u, v, fs=fs), t4(a, y, u, v, True, fs=fs)))
    for name, p, e in rows:
        print("%-44s probe %.4f ms   encode %.4f ms" % (name, p, e), flush=True)
    # a second layout: R and B planes in one group, G in another, all coded planes in the third
    lib.lumahip_exp_set_chan_stride(0)
    aba = None
    for k in range(1, n // 2):
        for i in range(n - 2 * k):
            x, y, z = i + 2 * k, i + k, i
            if va[y] - va[x] == va[z] - va[y] and gid[x] == gid[z] != gid[y] and not ({x, y, z} & {a, b, c}):
                aba = (x, y, z)
                break
        if aba:
            break
    if aba:
        x, y, z = aba
        third = [g for g in range(len(groups)) if g not in (gid[x], gid[y])]
        if third:
            pc = [i for i in range(n) if gid[i] == third[0] and i not in (x, y, z, a, b, c)]
            lib.lumahip_exp_set_chan_stride((va[y] - va[x]) // 4)
            print("%-44s probe %.4f ms   encode %.4f ms" % ("R, B in A, G in B; Y U V in C (one chunk)", t4(x, pc[0], pc[0], pc[0], fs=fs), t4(x, pc[0], pc[0], pc[0], True, fs=fs)))
            print("%-44s probe %.4f ms   encode %.4f ms" % ("R, B in A, G in B; Y in C, U V in C'", t4(x, pc[0], pc[1], pc[1], fs=fs), t4(x, pc[0], pc[1], pc[1], True, fs=fs)))
            lib.lumahip_exp_set_chan_stride(0)
    # decode: planes read, floats written
    pl = [chunks[ya].data_ptr() + offs[0], chunks[ub].data_ptr() + offs[1], chunks[uc].data_ptr() + offs[2]]
    ctx.encode_frames_device(va[a], fs, B, w, h, 1.0, profile, pl, st, psz)
    d_striped = ctx.time_launches(1, 3, va[a], fs, B, w, h, 1.0, profile, pl, st, psz)
    lib.lumahip_exp_set_chan_stride(0)
    d_packed = ctx.time_launches(1, 3, va[a], n3, B, w, h, 1.0, profile, [chunks[yb].data_ptr() + offs[0], chunks[uc].data_ptr() + offs[1], chunks[uc].data_ptr() + offs[2]], st, psz)
    print("decode: packed output frames in A (planes in B, C)   %.4f ms" % d_packed)
    print("decode: output R G B in A B C (planes in A, B, C)    %.4f ms" % d_striped)


if __name__ == "__main__":
    main()
