import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import lumahdrv_amd as L
w, h, B = 3840, 2160, 8
n3 = 3 * w * h
dev = torch.device("cuda:0")
src = torch.empty(4 * B * n3, dtype=torch.float32, device=dev)
for (name, ptf, bits, cs) in (("LINEAR-12 XYZ", L.PTF_LINEAR, 12, L.CS_XYZ), ("LINEAR-12 Luv", L.PTF_LINEAR, 12, L.CS_LUV), ("LINEAR-11 Luv", L.PTF_LINEAR, 11, L.CS_LUV),
                              ("PQ-12 Luv", L.PTF_PQ, 12, L.CS_LUV), ("PQ-13 Luv", L.PTF_PQ, 13, L.CS_LUV), ("PQ-14 Luv", L.PTF_PQ, 14, L.CS_LUV), ("PQ-16 Luv", L.PTF_PQ, 16, L.CS_LUV),
                              ("LOG-14 Luv", L.PTF_LOG, 14, L.CS_LUV)):
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(ptf, bits, cs, 8, 1e4, 0.005, L.build_lut(ptf, bits))
    info = ctx.quantizer_info()
    ctx.synth_frames_device(src.data_ptr(), n3, 4 * B, w, h)
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    planes = [torch.zeros(4 * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    res = []
    for d in (0, 1):
        ms = sorted(ctx.time_launches(d, 1, src.data_ptr() + (i % 4) * B * n3 * 4, n3, B, w, h, 1.0, 2,
                                      [planes[p].data_ptr() + (i % 4) * B * psz[p] for p in range(3)], st, psz) for i in range(9))[4]
        res.append(B * w * h / ms / 1e6)
    print("%-14s mode %s records %s keybits %s lds %s : encode %.1f decode %.1f Gpx/s" % (name, info["mode"], info.get("records"), info.get("key_bits"), info.get("lds_bytes"), res[0], res[1]), flush=True)
    ctx.close()
