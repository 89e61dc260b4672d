#!/usr/bin/env python3
"""tools/bench/numa_ab.py -- does it matter on which NUMA node the pinned staging rings and the copy threads of a context live?
Pageable 3840x2160 frames through lumahip_encode_frames_host / lumahip_decode_frames_host (the batched host entry points: staging
copies by the context's copy threads, 3-slot pipeline), for every combination of
  P = the node the CALLING thread runs on and its frames were first touched on (a worker thread of this script pins itself with
      sched_setaffinity; the process as a whole stays unrestricted),
  S = the node the staging rings / copy threads are placed on (lumahip_tune numa_node S), "off" (lumahip_tune numa 0: the
      runtime's default allocation, unpinned copy threads -- the behaviour before round 4), or "auto" (the GPU's own node).
One fresh context per combination and round; 11 rounds over all combinations in shuffled order (the first is a warm-up), 16 frames
per timed batch, output buffers allocated and touched outside the timed calls.  Median Mpixel/s.  -> profiles/r04_numa.txt"""
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd import capi  # noqa: E402


def cpus_of(node):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def on_node(node, fn):
    """run fn() on a thread pinned to `node`'s CPUs"""
    box = {}

    def body():
        os.sched_setaffinity(0, set(cpus_of(node)))     # pid 0 = the calling thread
        box["r"] = fn()
    t = threading.Thread(target=body)
    t.start()
    t.join()
    return box["r"]


ROUNDS = 11


def main():
    w, h, n = 3840, 2160, 16
    nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
    c0 = L.Context(0)
    print("host: %d NUMA nodes, %d CPUs usable; GPU 0: %s" % (len(nodes), len(os.sched_getaffinity(0)), c0.numa_info()))
    c0.close()
    lut = L.build_lut(L.PTF_PQ, 11, 1e4, 0.005)
    _, hs, st, _ = L.plane_geometry(w, h, 2)

    def make_buffers():
        rng = np.random.default_rng(1)
        fr = [np.exp(rng.uniform(np.log(1e-3), np.log(1e4), size=(3, h, w))).astype(np.float32) for _ in range(n)]
        pl = [[np.zeros((hs[p], st[p]), dtype=np.uint8) for p in range(3)] for _ in range(n)]
        out = [np.zeros((3, h, w), dtype=np.float32) for _ in range(n)]
        return fr, pl, out
    bufs = {P: on_node(P, make_buffers) for P in nodes}     # first touch on node P
    res = {}
    combos = [(P, S) for P in nodes for S in ["off", "auto"] + nodes]
    rnd = random.Random(4)
    for rep in range(ROUNDS):
        order = combos[:]
        rnd.shuffle(order)
        for (P, S) in order:
            fr, pl, out = bufs[P]

            def run():
                c = L.Context(0)
                c.tune("numa", 0 if S == "off" else 1)          # 1 = rings AND pinned copy threads (the library's default is rings only)
                if isinstance(S, int):
                    c.tune("numa_node", S)
                c.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, lut)
                fp = (C.c_void_p * n)(*[f.ctypes.data for f in fr])
                pp = (C.c_void_p * (3 * n))(*[p.ctypes.data for tri in pl for p in tri])
                op = (C.c_void_p * n)(*[o.ctypes.data for o in out])
                sa = (C.c_int * 3)(*st)
                means = (C.c_float * n)()
                te = td = 0.0
                for it in range(2):                                # the first pass allocates the rings and starts the threads
                    t0 = time.perf_counter()
                    c._chk(c.L.lumahip_encode_frames_host(c.h, fp, n, w, h, 1.0, 2, pp, sa, means))
                    te = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    c._chk(c.L.lumahip_decode_frames_host(c.h, pp, sa, n, w, h, 2, 1.0, op))
                    td = time.perf_counter() - t0
                info = c.numa_info()
                c.close()
                return te, td, info
            te, td, info = on_node(P, run)
            if rep > 0:
                res.setdefault((P, S), []).append((te, td, info))
    px = n * w * h / 1e6
    print("P (caller thread + frames) | S (staging rings + copy threads) | encode Mpixel/s | decode Mpixel/s | lumahip_numa_info")
    for (P, S) in combos:
        r = res[(P, S)]
        print("node %d                     | %-32s | %9.0f       | %9.0f       | %s" % (
            P, ("node %d" % S) if isinstance(S, int) else S, px / np.median([x[0] for x in r]), px / np.median([x[1] for x in r]), r[0][2]), flush=True)


if __name__ == "__main__":
    main()
