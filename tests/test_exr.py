"""ExrInterface (include/exr_interface.h) without OpenEXR: the C++ reader / writer against an independent
restatement of the OpenEXR scan-line layout written here in numpy + zlib.  CPU only."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lumahdrv_amd", "lib")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    import lumahdrv_amd
    lumahdrv_amd.build_library()
    exe = str(tmp_path_factory.mktemp("exr") / "exr_tool")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "exr_tool.cpp"), "-o", exe, "-L" + LIB, "-lluma_hip", "-llumahip",
                    "-Wl,-rpath," + LIB], check=True)
    return exe


def attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


# ---- PIZ and PXR24 ENCODERS, restated here from the published scheme (wavelet + LUT + Huffman with run-length symbol;
# ---- byte-plane delta + zlib).  Test infrastructure: the C++ reader has the decoders, nothing ships an encoder.

def _wenc14(a, b):
    as_ = a - 65536 if a >= 32768 else a
    bs = b - 65536 if b >= 32768 else b
    return ((as_ + bs) >> 1) & 0xFFFF, (as_ - bs) & 0xFFFF


def _wenc16(a, b):
    ao = (a + 0x8000) & 0xFFFF
    m = (ao + b) >> 1
    d = ao - b
    if d < 0:
        m = (m + 0x8000) & 0xFFFF
    return m, d & 0xFFFF


def _wav2_encode(a, base, nx, ox, ny, oy, mx):
    wenc = _wenc14 if mx < (1 << 14) else _wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        py, ey = base, base + oy * (ny - p2)
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i01 = wenc(a[px], a[p01])
                i10, i11 = wenc(a[p10], a[p11])
                a[px], a[p10] = wenc(i00, i10)
                a[p01], a[p11] = wenc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                i00, a[p10] = wenc(a[px], a[p10])
                a[px] = i00
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                i00, a[p01] = wenc(a[px], a[p01])
                a[px] = i00
                px += ox2
        p, p2 = p2, p2 << 1


class _BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0
        self.bits = 0

    def put(self, nbits, value):
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        self.bits += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def finish(self):
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 0xFF)
            self.acc = self.n = 0
        return bytes(self.out)


def _huf_compress_py(words):
    import heapq
    freq = {}
    for wv in words:
        freq[wv] = freq.get(wv, 0) + 1
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1                                        # the run-length pseudo-symbol
    # Huffman code lengths (any valid tree will do: the stream carries the lengths)
    heap = [(f, i, (s,)) for i, (s, f) in enumerate(sorted(freq.items()))]
    heapq.heapify(heap)
    length = {s: 0 for s in freq}
    cnt = len(heap)
    while len(heap) > 1:
        f1, _, s1 = heapq.heappop(heap)
        f2, _, s2 = heapq.heappop(heap)
        for x in s1 + s2:
            length[x] += 1
        heapq.heappush(heap, (f1 + f2, cnt, s1 + s2))
        cnt += 1
    assert max(length.values()) <= 58
    # canonical codes: longest codes first
    n = [0] * 59
    for l in length.values():
        n[l] += 1
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        n[i] = c
        c = nc
    code = {}
    for s_ in sorted(length):
        code[s_] = n[length[s_]]
        n[length[s_]] += 1
    # packed table: 6-bit lengths with zero-run escapes
    tb = _BitWriter()
    s_ = im
    while s_ <= iM:
        l = length.get(s_, 0)
        if l == 0:
            run = 1
            while s_ + run <= iM and run < 255 + 6 and length.get(s_ + run, 0) == 0:
                run += 1
            if run >= 2:
                if run >= 6:
                    tb.put(6, 63)
                    tb.put(8, run - 6)
                else:
                    tb.put(6, 59 + run - 2)
                s_ += run
                continue
        tb.put(6, l)
        s_ += 1
    table = tb.finish()
    # data with run-length coding where it is shorter
    db = _BitWriter()

    def send(sym, run):
        if run and length[sym] + length[iM] + 8 < length[sym] * run:
            db.put(length[sym], code[sym])
            db.put(length[iM], code[iM])
            db.put(8, run)
        else:
            for _ in range(run + 1):
                db.put(length[sym], code[sym])

    cur, run = words[0], 0
    for wv in words[1:]:
        if wv == cur and run < 255:
            run += 1
        else:
            send(cur, run)
            cur, run = wv, 0
    send(cur, run)
    nbits = db.bits
    data = db.finish()
    return struct.pack("<IIIII", im, iM, len(table), nbits, 0) + table + data


def piz_compress_py(chan_rows):
    """chan_rows: list (channel order) of 2-D arrays (rows of this block) -> PIZ chunk payload"""
    words, layout = [], []
    for a in chan_rows:
        w16 = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"))).view("<u2")     # (rows, width * size)
        size = a.dtype.itemsize // 2
        layout.append((len(words), a.shape[1], size, a.shape[0]))
        words.extend(int(x) for x in w16.reshape(-1))
    bitmap = bytearray(8192)
    for wv in set(words):
        bitmap[wv >> 3] |= 1 << (wv & 7)
    bitmap[0] &= ~1 & 0xFF
    nz = [i for i in range(8192) if bitmap[i]]
    min_nz, max_nz = (nz[0], nz[-1]) if nz else (8191, 0)
    lut, k = {}, 0
    for i in range(65536):
        if i == 0 or bitmap[i >> 3] & (1 << (i & 7)):
            lut[i] = k
            k += 1
    mx = k - 1
    words = [lut[wv] for wv in words]
    for start, nx, size, ny in layout:
        for j in range(size):
            _wav2_encode(words, start + j, nx, size, ny, nx * size, mx)
    huf = _huf_compress_py(words)
    out = struct.pack("<HH", min_nz, max_nz)
    if min_nz <= max_nz:
        out += bytes(bitmap[min_nz:max_nz + 1])
    return out + struct.pack("<i", len(huf)) + huf


def pxr24_compress_py(chan_rows):
    """byte planes of horizontally delta-coded samples, scan line by scan line, channel by channel; FLOAT as 24 bits"""
    out = bytearray()
    rows = chan_rows[0].shape[0]
    for y in range(rows):
        for a in chan_rows:
            if a.dtype == np.float16:
                v = a[y].view(np.uint16).astype(np.int64)
                d = np.diff(np.concatenate([[0], v])) & 0xFFFF
                planes = [(d >> 8) & 0xFF, d & 0xFF]
            elif a.dtype == np.float32:
                v = (a[y].view(np.uint32).astype(np.int64) >> 8)        # the caller passes values that fit 24 bits
                d = np.diff(np.concatenate([[0], v])) & 0xFFFFFF
                planes = [(d >> 16) & 0xFF, (d >> 8) & 0xFF, d & 0xFF]
            else:
                v = a[y].astype(np.int64)
                d = np.diff(np.concatenate([[0], v])) & 0xFFFFFFFF
                planes = [(d >> 24) & 0xFF, (d >> 16) & 0xFF, (d >> 8) & 0xFF, d & 0xFF]
            for pl in planes:
                out += pl.astype(np.uint8).tobytes()
    return zlib.compress(bytes(out))


def write_exr_py(path, chans, comp, x0=0, y0=0):
    """chans: dict name -> 2-D array (float16 / float32 / uint32).  comp: 0 none, 2 zips, 3 zip, 4 piz, 5 pxr24."""
    names = sorted(chans)
    h, w = chans[names[0]].shape
    tcode = {np.dtype("uint32"): 0, np.dtype("float16"): 1, np.dtype("float32"): 2}
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", tcode[chans[n].dtype], 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", x0, y0, x0 + w - 1, y0 + h - 1)
    hdr = struct.pack("<ii", 20000630, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp]))
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lpb = {3: 16, 4: 32, 5: 16}.get(comp, 1)
    nblk = (h + lpb - 1) // lpb
    chunks = []
    for b in range(nblk):
        rows = range(b * lpb, min(h, (b + 1) * lpb))
        raw = b"".join(chans[n][y].astype(chans[n].dtype.newbyteorder("<")).tobytes() for y in rows for n in names)
        if comp == 4:
            z = piz_compress_py([chans[n][rows.start:rows.stop] for n in names])
            data = z if len(z) < len(raw) else raw
        elif comp == 5:
            z = pxr24_compress_py([chans[n][rows.start:rows.stop] for n in names])
            data = z if len(z) < len(raw) else raw
        elif comp in (2, 3):
            a = np.frombuffer(raw, dtype=np.uint8)
            t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
            p = t.copy()
            p[1:] = (t[1:] - t[:-1] + 128) & 0xFF
            z = zlib.compress(p.astype(np.uint8).tobytes())
            data = z if len(z) < len(raw) else raw
        else:
            data = raw
        chunks.append(struct.pack("<ii", y0 + b * lpb, len(data)) + data)
    pos = len(hdr) + 8 * nblk
    table = b""
    for c in chunks:
        table += struct.pack("<Q", pos)
        pos += len(c)
    open(path, "wb").write(hdr + table + b"".join(chunks))


def read_exr_py(path):
    d = open(path, "rb").read()
    assert struct.unpack_from("<i", d, 0)[0] == 20000630
    p = 8
    info = {}
    while d[p] != 0:
        e = d.index(b"\0", p)
        name = d[p:e].decode()
        p = e + 1
        e = d.index(b"\0", p)
        p = e + 1
        size = struct.unpack_from("<i", d, p)[0]
        p += 4
        info[name] = d[p:p + size]
        p += size
    p += 1
    chans = []
    c = info["channels"]
    q = 0
    while c[q] != 0:
        e = c.index(b"\0", q)
        nm = c[q:e].decode()
        t = struct.unpack_from("<i", c, e + 1)[0]
        chans.append((nm, t))
        q = e + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<4i", info["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    comp = info["compression"][0]
    lpb = 16 if comp == 3 else 1
    nblk = (h + lpb - 1) // lpb
    offs = struct.unpack_from("<%dQ" % nblk, d, p)
    sz = {0: 4, 1: 2, 2: 4}
    dt = {0: "<u4", 1: "<f2", 2: "<f4"}
    line = sum(w * sz[t] for _, t in chans)
    out = {n: np.zeros((h, w), dtype=np.float32) for n, _ in chans}
    for o in offs:
        yy, n = struct.unpack_from("<ii", d, o)
        data = d[o + 8:o + 8 + n]
        lines = min(lpb, y1 - yy + 1)
        if comp in (2, 3) and n != line * lines:
            t = np.frombuffer(zlib.decompress(data), dtype=np.uint8).astype(np.int32)
            t = (np.cumsum(t - 128) + 128) & 0xFF   # inverse predictor
            t = t.astype(np.uint8)
            half = (t.size + 1) // 2
            a = np.empty(t.size, dtype=np.uint8)
            a[0::2] = t[:half]
            a[1::2] = t[half:]
            data = a.tobytes()
        q = 0
        for l in range(lines):
            for nm, tt in chans:
                out[nm][yy - y0 + l] = np.frombuffer(data, dtype=dt[tt], count=w, offset=q).astype(np.float32)
                q += w * sz[tt]
    return out, comp


def cpp_read(tool, path, tmp):
    outp = str(tmp / "out.f32")
    subprocess.run([tool, "read", path, outp], check=True)
    raw = open(outp, "rb").read()
    w, h = struct.unpack_from("<II", raw, 0)
    return np.frombuffer(raw, dtype="<f4", offset=8).reshape(3, h, w)


def eq(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_half_conversion_matches_ieee(tool):
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 1e3,
                        np.exp(rng.uniform(-25, 12, 20000)).astype(np.float32),
                        np.array([0, -0.0, 65504, 65519.99, 65520, 1e9, -1e9, 6e-8, 5.9e-8, 2.98e-8, 2.99e-8, 6.1e-5, np.inf,
                                  -np.inf, np.nan, 1.0009765625, 1.00048828125, 1.00146484375], dtype=np.float32)])
    r = subprocess.run([tool, "half"], input=v.tobytes(), capture_output=True, check=True)
    got = np.frombuffer(r.stdout, dtype="<u2")
    with np.errstate(over="ignore"):
        exp = v.astype(np.float16).view(np.uint16)
    nan = np.isnan(v)
    assert np.array_equal(got[~nan], exp[~nan])
    assert np.all((got[nan] & 0x7C00) == 0x7C00) and np.all((got[nan] & 0x3FF) != 0)


@pytest.mark.parametrize("comp", [0, 2, 3])
@pytest.mark.parametrize("dtype", ["float16", "float32"])
def test_reader_against_python_writer(tool, tmp_path, comp, dtype):
    rng = np.random.default_rng(comp * 7 + len(dtype))
    h, w = 37, 50
    ch = {n: np.exp(rng.uniform(-8, 11.5, (h, w))).astype(dtype) for n in "RGBA"}
    path = str(tmp_path / "in.exr")
    write_exr_py(path, ch, comp, x0=-3, y0=5)
    got = cpp_read(tool, path, tmp_path)
    with np.errstate(over="ignore"):
        for i, n in enumerate("RGB"):
            assert eq(got[i], ch[n].astype(np.float16).astype(np.float32)), n   # Imf::Rgba: everything through half


def test_single_channel_replication_and_errors(tool, tmp_path):
    h, w = 8, 6
    g = np.linspace(0.1, 900, h * w).reshape(h, w).astype(np.float16)
    p = str(tmp_path / "g.exr")
    write_exr_py(p, {"G": g}, 3)
    got = cpp_read(tool, p, tmp_path)
    assert all(eq(got[i], g.astype(np.float32)) for i in range(3))     # WRITE_G: replicated to all three planes
    p2 = str(tmp_path / "y.exr")
    write_exr_py(p2, {"Y": g}, 0)
    r = subprocess.run([tool, "read", p2, str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 1 and "luminance only" in r.stderr             # src/exr_interface.cpp:145
    open(str(tmp_path / "junk.exr"), "wb").write(b"not an exr file at all")
    r = subprocess.run([tool, "read", str(tmp_path / "junk.exr"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 1


@pytest.mark.parametrize("comp,as_float", [(0, 0), (2, 0), (3, 0), (3, 1)])
def test_writer_against_python_reader_and_roundtrip(tool, tmp_path, comp, as_float):
    rng = np.random.default_rng(11)
    h, w = 40, 33
    f = np.exp(rng.uniform(-6, 11, (3, h, w))).astype(np.float32)
    f[0, 0, :4] = [0, 70000, -2.5, 1e-9]
    src = str(tmp_path / "src.f32")
    open(src, "wb").write(struct.pack("<II", w, h) + f.tobytes())
    out = str(tmp_path / "w.exr")
    subprocess.run([tool, "write", src, out, str(comp), str(as_float)], check=True)
    chans, c = read_exr_py(out)
    assert c == comp and sorted(chans) == ["B", "G", "R"]
    with np.errstate(over="ignore"):
        exp = f if as_float else f.astype(np.float16).astype(np.float32)
        for i, n in enumerate("RGB"):
            assert eq(chans[n], exp[i])
        back = cpp_read(tool, out, tmp_path)
        assert eq(back, f.astype(np.float16).astype(np.float32))            # reading narrows to half either way


def test_reader_survives_corrupted_files_under_sanitizers(tmp_path):
    """mutation fuzzing (truncation, bit flips, stomped 32- and 64-bit fields, random spans) of valid files of every
    compression the reader implements (NONE, RLE, ZIPS, ZIP from the writer; PIZ and PXR24 from the Python encoders),
    reader built with ASan + UBSan: every attempt either decodes or raises LumaException"""
    exe = str(tmp_path / "exr_fuzz")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "exr_fuzz.cpp"),
                    os.path.join(ROOT, "lumahdrv_amd", "csrc", "facade", "exr_interface.cpp"),
                    os.path.join(ROOT, "lumahdrv_amd", "csrc", "facade", "exr_codecs.cpp"), "-o", exe, "-lz"], check=True)
    # seeds for the compressions only the reader implements
    rng = np.random.default_rng(8)
    smooth = (np.add.outer(np.arange(40), np.arange(29)) // 5).astype(np.float16)
    seeds = []
    for comp, nm in ((4, "piz.exr"), (5, "pxr24.exr")):
        sp = str(tmp_path / nm)
        write_exr_py(sp, {"R": smooth, "G": rng.uniform(0, 9, smooth.shape).astype(np.float16) // 1, "B": (smooth * 3).astype(np.float16)}, comp)
        seeds.append(sp)
    r = subprocess.run([exe, str(tmp_path), "400"] + seeds, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.startswith("ok ")
    d, rj = [int(x.split("=")[1]) for x in r.stdout.split()[1:3]]
    assert d + rj == 2400 and rj > 300


def test_wrapping_chunk_offsets_and_empty_attributes_are_rejected(tmp_path):
    """Directed cases from the round-1 review, reader + CLI built with ASan + UBSan (no GPU library needed: exr_tool links
    the reader source directly): a 64-bit chunk offset that wraps `p + n` (0xFFFFFFFFFFFFFFFE), offsets at / past the end
    of the file, and a zero-size `compression` / `lineOrder` attribute as the last bytes of a truncated file."""
    exe = str(tmp_path / "exr_tool_asan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "exr_tool.cpp"),
                    os.path.join(ROOT, "lumahdrv_amd", "csrc", "facade", "exr_interface.cpp"),
                    os.path.join(ROOT, "lumahdrv_amd", "csrc", "facade", "exr_codecs.cpp"), "-o", exe, "-lz"], check=True)
    rng = np.random.default_rng(4)
    chans = {n: rng.uniform(0, 100, (18, 20)).astype(np.float16) for n in "RGB"}
    for comp in (0, 2, 3):
        good = str(tmp_path / ("good%d.exr" % comp))
        write_exr_py(good, chans, comp)
        d = bytearray(open(good, "rb").read())
        p = 8
        while d[p] != 0:                                  # walk the attributes to the offset table
            p = d.index(b"\0", p) + 1
            p = d.index(b"\0", p) + 1
            p += 4 + struct.unpack_from("<i", d, p)[0]
        table = p + 1
        r = subprocess.run([exe, "read", good, str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for v in (0xFFFFFFFFFFFFFFFE, 0xFFFFFFFFFFFFFFF8, 0x8000000000000000, len(d), len(d) - 1, len(d) - 7, 1 << 32):
            m = bytearray(d)
            struct.pack_into("<Q", m, table, v)
            bad = str(tmp_path / "bad.exr")
            open(bad, "wb").write(m)
            r = subprocess.run([exe, "read", bad, str(tmp_path / "o.bin")], capture_output=True, text=True)
            assert r.returncode == 1 and "LumaException" in r.stderr, (comp, hex(v), r.returncode, r.stderr[-600:])
    # zero-size compression / lineOrder attribute at the very end of the data
    for name, typ in (("compression", "compression"), ("lineOrder", "lineOrder")):
        hdr = struct.pack("<ii", 20000630, 2) + attr(name, typ, b"")
        bad = str(tmp_path / "short.exr")
        open(bad, "wb").write(hdr)
        r = subprocess.run([exe, "read", bad, str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 1 and "LumaException" in r.stderr, (name, r.returncode, r.stderr[-600:])


@pytest.mark.parametrize("comp", [4, 5])
def test_reader_decodes_piz_and_pxr24(tool, tmp_path, comp):
    """PIZ (wavelet + LUT + Huffman) and PXR24 (24-bit byte-plane delta + zlib) chunks written by the Python encoders
    above: odd sizes across block boundaries, HALF and FLOAT channels, few distinct values (14-bit wavelet path and
    Huffman run-length codes) and noise (16-bit wavelet path), an alpha channel, a single-channel file, a data window
    that does not start at the origin."""
    rng = np.random.default_rng(comp)
    cases = []
    h, w = 45, 37                                            # two PIZ blocks (32 + 13), three PXR24 blocks
    smooth = (np.add.outer(np.arange(h), np.arange(w)) // 7).astype(np.float16)
    cases.append(({"R": smooth, "G": (smooth * 2).astype(np.float16), "B": np.zeros((h, w), np.float16)}, "smooth half"))
    noise = {n: rng.uniform(0, 6e4, (70, 300)).astype(np.float16) for n in "RGBA"}       # > 16384 distinct words per block
    cases.append((noise, "noise half + alpha"))
    # a ramp through > 16384 distinct half codes per block that still compresses: the 16-bit wavelet path of the decoder
    idx = np.arange(64 * 300, dtype=np.int64).reshape(64, 300) % (32 * 300)
    ramp = {"R": (3 * idx).astype(np.uint16).view(np.float16), "G": (3 * idx + 1).astype(np.uint16).view(np.float16),
            "B": (32768 + 3 * idx + 2).astype(np.uint16).view(np.float16)}
    assert all(np.isfinite(v.astype(np.float32)).all() for v in ramp.values())
    cases.append((ramp, "ramp, 16-bit wavelet"))
    f24 = {n: (rng.uniform(1e-3, 1e4, (h, w)).astype(np.float32).view(np.uint32) & 0xFFFFFF00).view(np.float32) for n in "RGB"}
    cases.append((f24, "float"))
    mixed = {"R": rng.uniform(0, 100, (33, 9)).astype(np.float16), "G": f24["G"][:33, :9].copy(), "B": rng.uniform(0, 5, (33, 9)).astype(np.float16)}
    cases.append((mixed, "mixed half / float"))
    cases.append(({"G": rng.uniform(0, 50, (5, 3)).astype(np.float16)}, "single channel, tiny"))
    for chans, what in cases:
        p = str(tmp_path / "c.exr")
        write_exr_py(p, chans, comp, x0=3, y0=-2)
        got = cpp_read(tool, p, tmp_path)
        names = [n for n in "RGB" if n in chans]
        src = names if len(names) == 3 else [names[0]] * 3
        with np.errstate(over="ignore"):
            for i, n in enumerate(src):
                assert eq(got[i], chans[n].astype(np.float16).astype(np.float32)), (comp, what, n)
    # the compressed path was really taken (not the stored-raw fallback) for the compressible cases
    write_exr_py(str(tmp_path / "s.exr"), cases[0][0], comp)
    assert os.path.getsize(str(tmp_path / "s.exr")) < 45 * 37 * 6 // 2
    if comp == 4:
        write_exr_py(str(tmp_path / "r.exr"), ramp, comp)
        assert os.path.getsize(str(tmp_path / "r.exr")) < 64 * 300 * 6 * 0.8
        assert len({int(x) for v in ramp.values() for x in v[:32].view(np.uint16).reshape(-1)}) > 16384


def test_reader_on_a_file_written_by_openexr(tool, tmp_path):
    """tests/golden/openexr_written_16x16_rgba_half.exr is a genuine OpenEXR-library file (CPython's test-suite sample
    `imghdrdata/python.exr`, 16x16 RGBA HALF, uncompressed, increasing-y) -- the one real-world EXR in the build image.
    The C++ reader and the Python restatement of the layout must agree on it."""
    path = os.path.join(ROOT, "tests", "golden", "openexr_written_16x16_rgba_half.exr")
    ch, comp = read_exr_py(path)
    assert comp == 0 and sorted(ch) == ["A", "B", "G", "R"] and ch["R"].shape == (16, 16)
    got = cpp_read(tool, path, tmp_path)
    for i, n in enumerate("RGB"):
        assert eq(got[i], ch[n])
    assert float(np.max(got)) > 0.0
