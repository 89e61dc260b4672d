#!/usr/bin/env python3
"""tools/bench/ab_inproc.py <libA.so> <libB.so> [workload] [rounds] -- in-process, interleaved A/B of two builds of liblumahip.so.

Separate processes cannot resolve differences below ~4 %: the traffic-only probe (identical code in every build) moves
by that much from process to process on one box.  Here both libraries are loaded into ONE process (RTLD_LOCAL |
RTLD_DEEPBIND, so that neither binds to the other's symbols), share the same device buffers and alternate
A, B, A, B ... launch groups; every group is 25 launches walking the 500-frame resident stream.  Prints the median and
the quartiles of the per-launch time of each build and of the paired differences."""
import ctypes
import importlib.util
import os
import statistics
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def load(path, name):
    os.environ["LUMAHIP_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "lumahdrv_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    ns = types.SimpleNamespace(**{k: getattr(ctypes, k) for k in dir(ctypes)})
    ns.CDLL = lambda p, mode=0: ctypes.CDLL(p, mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND | os.RTLD_NOW)
    m.C = ns
    m.lib()
    return m


def main():
    a_path, b_path = sys.argv[1], sys.argv[2]
    wl = sys.argv[3] if len(sys.argv) > 3 else "pq11_luv"
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 150
    direction = int(os.environ.get("AB_DIRECTION", "0"))   # 0 encode, 1 decode
    mods = [load(a_path, "capi_a"), load(b_path, "capi_b")]
    cfgs = {"pq11_luv": (1, 11, 0, 8, 1e4, 0.005, 1.0), "pq10_ycbcr": (1, 10, 2, 10, 1000.0, 0.01, 20.0),
            "log12_luv": (2, 12, 0, 8, 1e4, 0.005, 1.0), "pq11_rgb": (1, 11, 1, 8, 1e4, 0.005, 1.0), "pq13_luv": (1, 13, 0, 8, 1e4, 0.005, 1.0),
            "pq8_luv": (1, 8, 0, 8, 1e4, 0.005, 1.0)}
    ptf, bits, cs, bitsC, mx, mn, sc = cfgs[wl]
    w, h, B, nb, profile = 3840, 2160, int(os.environ.get("AB_FRAMES", "20")), int(os.environ.get("AB_BATCHES", "25")), int(os.environ.get("AB_PROFILE", "2"))
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = mods[0].plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctxs = []
    for side, m in zip("AB", mods):
        # AB_ENV_A / AB_ENV_B = "NAME=value,NAME=value": environment switches read when that side's context is created
        extra = dict(kv.split("=", 1) for kv in os.environ.get("AB_ENV_" + side, "").split(",") if "=" in kv)
        os.environ.update(extra)
        os.environ["LUMAHIP_TUNING"] = "1"   # the LUMAHIP_* overrides are honoured only under this gate
        c = m.Context(0)
        for k in extra:
            os.environ.pop(k, None)
        c.set_quantizer(ptf, bits, cs, bitsC, mx, mn, m.build_lut(ptf, bits, mx, mn))
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        ctxs.append(c)
    for b in range(nb):
        ctxs[0].synth_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, 20250929, b * B)
    torch.cuda.synchronize()

    iters = int(os.environ.get("AB_ITERS", "1"))   # > 1: that many back-to-back launches per batch (sustained clocks)
    state = {"b": 0}

    def group(c):
        if iters > 1:
            b = state["b"] = (state["b"] + 1) % nb
            pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
            return c.time_launches(direction, iters, src.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, profile, pl, st, psz)
        t = 0.0
        for b in range(nb):
            pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
            t += c.time_launches(direction, 1, src.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, profile, pl, st, psz)
        return t / nb

    if direction == 1:   # decode needs planes: produce them once
        for b in range(nb):
            pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
            ctxs[0].encode_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, profile, pl, st, psz)
        torch.cuda.synchronize()
    for _ in range(3):
        group(ctxs[0]); group(ctxs[1])
    ta, tb = [], []
    for r in range(rounds):
        order = (0, 1) if r % 2 == 0 else (1, 0)
        res = {}
        for i in order:
            res[i] = group(ctxs[i])
        ta.append(res[0]); tb.append(res[1])
    d = [y - x for x, y in zip(ta, tb)]
    q = lambda v: statistics.quantiles(v, n=4)
    print("%s  A = %s  B = %s  (%d rounds of 25 launches each, %s)" % (wl, a_path, b_path, rounds, "decode" if direction else "encode"))
    print("A  median %.4f ms  quartiles %.4f / %.4f" % (statistics.median(ta), q(ta)[0], q(ta)[2]))
    print("B  median %.4f ms  quartiles %.4f / %.4f" % (statistics.median(tb), q(tb)[0], q(tb)[2]))
    print("B - A  median %+.4f ms (%+.2f %%)  quartiles %+.4f / %+.4f" % (statistics.median(d), 100 * statistics.median(d) / statistics.median(ta), q(d)[0], q(d)[2]))


if __name__ == "__main__":
    main()
