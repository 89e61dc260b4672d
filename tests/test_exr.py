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


def write_exr_py(path, chans, comp, x0=0, y0=0):
    """chans: dict name -> 2-D array (float16 / float32 / uint32).  comp: 0 none, 2 zips, 3 zip."""
    names = sorted(chans)
    h, w = chans[names[0]].shape
    tcode = {np.dtype("uint32"): 0, np.dtype("float16"): 1, np.dtype("float32"): 2}
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", tcode[chans[n].dtype], 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", x0, y0, x0 + w - 1, y0 + h - 1)
    hdr = struct.pack("<ii", 20000630, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp]))
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lpb = 16 if comp == 3 else 1
    nblk = (h + lpb - 1) // lpb
    chunks = []
    for b in range(nblk):
        rows = range(b * lpb, min(h, (b + 1) * lpb))
        raw = b"".join(chans[n][y].astype(chans[n].dtype.newbyteorder("<")).tobytes() for y in rows for n in names)
        if comp in (2, 3):
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
    """mutation fuzzing (truncation, bit flips, stomped size fields, random spans) of valid files written with each
    compression, reader built with ASan + UBSan: every attempt either decodes or raises LumaException"""
    exe = str(tmp_path / "exr_fuzz")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "exr_fuzz.cpp"),
                    os.path.join(ROOT, "lumahdrv_amd", "csrc", "facade", "exr_interface.cpp"), "-o", exe, "-lz"], check=True)
    r = subprocess.run([exe, str(tmp_path), "400"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.startswith("ok ")
    d, rj = [int(x.split("=")[1]) for x in r.stdout.split()[1:3]]
    assert d + rj == 1600 and rj > 200


def test_wrapping_chunk_offsets_and_empty_attributes_are_rejected(tmp_path):
    """Directed cases from the round-1 review, reader + CLI built with ASan + UBSan (no GPU library needed: exr_tool links
    the reader source directly): a 64-bit chunk offset that wraps `p + n` (0xFFFFFFFFFFFFFFFE), offsets at / past the end
    of the file, and a zero-size `compression` / `lineOrder` attribute as the last bytes of a truncated file."""
    exe = str(tmp_path / "exr_tool_asan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "exr_tool.cpp"),
                    os.path.join(ROOT, "lumahdrv_amd", "csrc", "facade", "exr_interface.cpp"), "-o", exe, "-lz"], check=True)
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
