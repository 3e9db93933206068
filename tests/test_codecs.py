"""CPU suite, part 4: the PNG / EXR codecs behind mid_image_load / mid_image_save.

The reference decodes with lodepng and tinyexr (un-vendored submodules, absent here), so there are
no reference fixtures; PNG is cross-checked against an independent implementation that IS in this
image (Pillow), in both directions and over every colour type / bit depth / interlace mode; EXR
(no second implementation available) is checked by round trips, by hand-built files that exercise
each compression / pixel type the reader accepts, and by byte-level checks of the header."""
import io
import struct
import zlib

import numpy as np
import pytest

import image_denoising_filter_amd as mid

PIL = pytest.importorskip("PIL.Image")


def test_png_roundtrip_and_pillow_reads_ours(tmp_path):
    rng = np.random.default_rng(0)
    for shape in ((1, 1), (7, 13), (64, 64), (33, 200)):
        a = rng.integers(0, 256, (*shape, 4), dtype=np.uint8)
        p = tmp_path / f"a_{shape[0]}x{shape[1]}.png"
        mid.save_image(p, a)
        assert np.array_equal(mid.load_image(p), a)
        assert np.array_equal(np.asarray(PIL.open(p).convert("RGBA")), a)


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P", "1", "I;16"])
@pytest.mark.parametrize("interlace", [False, True])
def test_png_we_read_pillow_files(tmp_path, mode, interlace):
    rng = np.random.default_rng(1)
    h, w = 19, 23
    if mode == "I;16":
        im = PIL.fromarray(rng.integers(0, 65536, (h, w), dtype=np.uint16))
        expect = np.asarray(im, dtype=np.uint16) >> 8                       # lodepng keeps the most significant byte
        expect = np.stack([expect] * 3 + [np.full((h, w), 255)], -1).astype(np.uint8)
    else:
        im = PIL.fromarray(rng.integers(0, 256, (h, w, 4), dtype=np.uint8), "RGBA").convert(mode)
        expect = np.asarray(im.convert("RGBA"))
    p = tmp_path / "x.png"
    if interlace:
        # Pillow cannot write Adam7; build the interlaced stream by hand from the raw rows
        raw = _png_rows(im)
        _write_png(p, im, raw, interlace=True)
        if mode == "P":
            expect = expect.copy()
            expect[..., 3] = 255                               # the hand-built file carries no tRNS chunk
    else:
        im.save(p, optimize=(mode == "P"))
    assert np.array_equal(mid.load_image(p), expect)


def _png_rows(im):
    """(colour type, bit depth, bytes per pixel or bits, rows as bytes per pixel list)"""
    mode = im.mode
    arr = np.asarray(im)
    if mode == "1":
        return 0, 1, (arr.astype(np.uint8))
    if mode == "I;16":
        return 0, 16, arr.astype(">u2")
    return {"L": 0, "RGB": 2, "P": 3, "LA": 4, "RGBA": 6}[mode], 8, arr


def _write_png(path, im, raw, interlace):
    ctype, depth, arr = raw
    h, w = arr.shape[:2]

    def pack_rows(sub):
        out = b""
        for row in sub:
            if depth == 1:
                bits = np.packbits(row.astype(np.uint8))
                out += b"\0" + bits.tobytes()
            else:
                out += b"\0" + np.ascontiguousarray(row).tobytes()
        return out
    data = b""
    if interlace:
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = arr[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                data += pack_rows(sub)
    else:
        data = pack_rows(arr)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if ctype == 3:
        pal = im.getpalette()
        out += chunk(b"PLTE", bytes(pal[:768]))
    out += chunk(b"IDAT", zlib.compress(data)) + chunk(b"IEND", b"")
    open(path, "wb").write(out)


def test_png_palette_transparency_and_filters(tmp_path):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (31, 37, 4), dtype=np.uint8)
    im = PIL.fromarray(a, "RGBA")
    # Pillow picks per-row filters with compress_level > 0: all five filter types get exercised
    p = tmp_path / "f.png"
    im.save(p, compress_level=9)
    assert np.array_equal(mid.load_image(p), a)
    # palette + tRNS
    pim = im.convert("P", palette=PIL.ADAPTIVE, colors=17)
    pim.info["transparency"] = bytes([0, 128, 255] + [255] * 14)
    p2 = tmp_path / "p.png"
    pim.save(p2, transparency=bytes([0, 128, 255] + [255] * 14))
    assert np.array_equal(mid.load_image(p2), np.asarray(PIL.open(p2).convert("RGBA")))


def test_png_errors(tmp_path):
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not a png at all")
    with pytest.raises(mid.MidError) as e:
        mid.load_image(bad)
    assert e.value.code == 5
    with pytest.raises(mid.MidError):
        mid.load_image(tmp_path / "missing.png")
    good = tmp_path / "g.png"
    mid.save_image(good, np.zeros((4, 4, 4), np.uint8))
    blob = bytearray(good.read_bytes())
    blob[40] ^= 0xff                                        # corrupt IDAT -> CRC mismatch
    bad.write_bytes(bytes(blob))
    with pytest.raises(mid.MidError):
        mid.load_image(bad)


# ---- EXR ------------------------------------------------------------------------------------------
def _exr(w, h, channels, compression, lines, pixel_bytes, data_window=None, line_order=0):
    """Hand-built scanline EXR: channels = [(name, type)], lines[y] = {name: bytes}."""
    dw = data_window or (0, 0, w - 1, h - 1)

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", t, 0, 0, 0, 0, 1, 1) for n, t in channels) + b"\0"
    hdr = struct.pack("<ii", 20000630, 2)
    hdr += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    hdr += attr("dataWindow", "box2i", struct.pack("<4i", *dw)) + attr("displayWindow", "box2i", struct.pack("<4i", *dw))
    hdr += attr("lineOrder", "lineOrder", bytes([line_order])) + attr("pixelAspectRatio", "float", struct.pack("<f", 1))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1))
    hdr += b"\0"
    lpb = 16 if compression == 3 else 32 if compression == 4 else 1
    blocks = []
    for y0 in range(0, h, lpb):
        raw = b"".join(lines[y][n] for y in range(y0, min(h, y0 + lpb)) for n, _ in channels)
        if compression == 4:
            import piz_encoder
            blocks.append((dw[1] + y0, piz_encoder.piz_block(lines[y0:min(h, y0 + lpb)], channels, w, use_runs=pixel_bytes != "noruns")))
            continue
        if compression in (2, 3):
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]])
            d = t.astype(np.int16)
            d[1:] = (d[1:] - d[:-1] + 128) & 255
            z = zlib.compress(d.astype(np.uint8).tobytes())
            payload = z if len(z) < len(raw) else raw
        elif compression == 4:
            payload = None                      # filled per 32-line block below
        elif compression == 1:
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]])
            d = t.astype(np.int16)
            d[1:] = (d[1:] - d[:-1] + 128) & 255
            d = d.astype(np.uint8).tobytes()
            payload = b"".join(bytes([256 - 1 & 0xff]) + d[i:i + 1] for i in range(len(d)))   # all literals (count -1)
            if len(payload) >= len(raw):
                payload = raw
        else:
            payload = raw
        blocks.append((dw[1] + y0, payload))
    if line_order == 1:
        order = list(reversed(range(len(blocks))))
    else:
        order = list(range(len(blocks)))
    table_at = len(hdr)
    pos = table_at + 8 * len(blocks)
    offsets = [0] * len(blocks)
    body = b""
    for i in order:
        offsets[i] = pos + len(body)
        y, payload = blocks[i]
        body += struct.pack("<ii", y, len(payload)) + payload
    return hdr + b"".join(struct.pack("<Q", o) for o in offsets) + body


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
@pytest.mark.parametrize("ptype", [1, 2])
def test_exr_reader_handbuilt(tmp_path, compression, ptype):
    rng = np.random.default_rng(compression * 3 + ptype)
    h, w = 37, 21
    px = (rng.random((h, w, 4)) * 4).astype(np.float16 if ptype == 1 else np.float32)
    chans = [("A", ptype), ("B", ptype), ("G", ptype), ("R", ptype)]
    idx = {"R": 0, "G": 1, "B": 2, "A": 3}
    lines = [{n: np.ascontiguousarray(px[y, :, idx[n]]).tobytes() for n, _ in chans} for y in range(h)]
    p = tmp_path / "t.exr"
    p.write_bytes(_exr(w, h, chans, compression, lines, None, data_window=(5, -3, 5 + w - 1, -3 + h - 1),
                       line_order=compression % 2))
    got = mid.load_image(p)
    assert got.dtype == np.float32 and got.shape == (h, w, 4)
    assert np.array_equal(got, px.astype(np.float32))


def test_exr_missing_alpha_gray_and_uint(tmp_path):
    h, w = 5, 9
    rng = np.random.default_rng(9)
    rgb = rng.random((h, w, 3)).astype(np.float32)
    chans = [("B", 2), ("G", 2), ("R", 2)]
    lines = [{"R": rgb[y, :, 0].tobytes(), "G": rgb[y, :, 1].tobytes(), "B": rgb[y, :, 2].tobytes()} for y in range(h)]
    p = tmp_path / "rgb.exr"
    p.write_bytes(_exr(w, h, chans, 2, lines, None))
    got = mid.load_image(p)
    assert np.array_equal(got[..., :3], rgb) and np.all(got[..., 3] == 1.0)       # alpha defaults to 1 (README.md:59)
    y_ = rng.integers(0, 1000, (h, w)).astype(np.uint32)
    p2 = tmp_path / "y.exr"
    p2.write_bytes(_exr(w, h, [("Y", 0)], 0, [{"Y": y_[r].tobytes()} for r in range(h)], None))
    g = mid.load_image(p2)
    assert np.array_equal(g[..., 0], y_.astype(np.float32)) and np.array_equal(g[..., 0], g[..., 2])
    # half special values
    hv = np.array([0x0000, 0x8000, 0x0001, 0x03ff, 0x0400, 0x7bff, 0x7c00, 0xfc00, 0x3c00], np.uint16)
    p3 = tmp_path / "h.exr"
    p3.write_bytes(_exr(9, 1, [("Y", 1)], 0, [{"Y": hv.tobytes()}], None))
    assert np.array_equal(mid.load_image(p3)[0, :, 0], hv.view(np.float16).astype(np.float32))


def test_exr_write_roundtrip_and_header(tmp_path):
    rng = np.random.default_rng(4)
    for shape in ((3, 5), (16, 16), (40, 33), (17, 64)):
        a = (rng.standard_normal((*shape, 4)) * 3).astype(np.float32)
        a[0, 0] = [np.inf, -0.0, 1e-42, 65504.0]
        p = tmp_path / f"r{shape[0]}.exr"
        mid.save_image(p, a)
        b = mid.load_image(p)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "float EXR round trip must be bit-exact"
        blob = p.read_bytes()
        assert struct.unpack("<ii", blob[:8]) == (20000630, 2)
        assert blob[8:17] == b"channels\0"
        names = [n for n in (b"A\0", b"B\0", b"G\0", b"R\0")]
        pos = blob.index(b"chlist\0") + 7 + 4
        for n in names:                                       # A,B,G,R as FLOAT, like SaveEXR(...,4,0)
            assert blob[pos:pos + 2] == n and struct.unpack("<i", blob[pos + 2:pos + 6])[0] == 2
            pos += 2 + 16
        comp = blob[blob.index(b"compression\0compression\0") + 24 + 4]
        assert comp == (0 if shape[0] < 16 and shape[1] < 16 else 3)


def test_exr_unsupported_features_are_named(tmp_path):
    a = np.zeros((4, 4, 4), np.float32)
    p = tmp_path / "a.exr"
    mid.save_image(p, a)
    blob = bytearray(p.read_bytes())
    i = blob.index(b"compression\0compression\0") + 24 + 4
    blob[i] = 6                                               # B44
    (tmp_path / "b44.exr").write_bytes(bytes(blob))
    with pytest.raises(mid.MidError) as e:
        mid.load_image(tmp_path / "b44.exr")
    assert "B44" in str(e.value) and e.value.code == 5
    blob2 = bytearray(p.read_bytes())
    blob2[5] |= 0x02                                          # tiled bit without a tiles attribute
    (tmp_path / "tiled.exr").write_bytes(bytes(blob2))
    with pytest.raises(mid.MidError) as e:
        mid.load_image(tmp_path / "tiled.exr")
    assert "tiles attribute" in str(e.value)
    mip = _exr_tiled(8, 8, [("R", 2)], 0, np.zeros((8, 8, 1), np.float32), 4, 4, level_mode=1)
    (tmp_path / "mip.exr").write_bytes(mip)
    with pytest.raises(mid.MidError) as e:
        mid.load_image(tmp_path / "mip.exr")
    assert "ONE_LEVEL" in str(e.value)


def _zip_block(raw):
    t = np.frombuffer(raw, np.uint8)
    t = np.concatenate([t[0::2], t[1::2]])
    d = t.astype(np.int16)
    d[1:] = (d[1:] - d[:-1] + 128) & 255
    z = zlib.compress(d.astype(np.uint8).tobytes())
    return z if len(z) < len(raw) else raw


def _exr_tiled(w, h, channels, compression, px, tw, th, level_mode=0, shuffle=None):
    """Hand-built single-level tiled EXR (independent of the reader): px[y, x, c] in channel order, NONE or ZIP tiles;
    tile chunks = tileX, tileY, levelX, levelY, size, data (per line: channel after channel)."""
    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", t, 0, 0, 0, 0, 1, 1) for n, t in channels) + b"\0"
    dw = (0, 0, w - 1, h - 1)
    hdr = struct.pack("<ii", 20000630, 2 | 0x200)
    hdr += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    hdr += attr("dataWindow", "box2i", struct.pack("<4i", *dw)) + attr("displayWindow", "box2i", struct.pack("<4i", *dw))
    hdr += attr("lineOrder", "lineOrder", bytes([0])) + attr("pixelAspectRatio", "float", struct.pack("<f", 1))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1))
    hdr += attr("tiles", "tiledesc", struct.pack("<IIB", tw, th, level_mode)) + b"\0"
    tiles = []
    for ty in range((h + th - 1) // th):
        for tx in range((w + tw - 1) // tw):
            sub = px[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            raw = b"".join(np.ascontiguousarray(sub[y, :, c]).tobytes() for y in range(sub.shape[0]) for c in range(len(channels)))
            tiles.append((tx, ty, _zip_block(raw) if compression == 3 else raw))
    order = list(range(len(tiles)))
    if shuffle is not None:
        shuffle.shuffle(order)                                # tiles may sit in the file in any order
    pos = len(hdr) + 8 * len(tiles)
    offsets, body = [0] * len(tiles), b""
    for i in order:
        offsets[i] = pos + len(body)
        tx, ty, payload = tiles[i]
        body += struct.pack("<iiiii", tx, ty, 0, 0, len(payload)) + payload
    return hdr + b"".join(struct.pack("<Q", o) for o in offsets) + body


@pytest.mark.parametrize("compression", [0, 3])
@pytest.mark.parametrize("ptype,tile", [(2, (16, 16)), (1, (32, 8)), (2, (64, 64))])
def test_exr_tiled_one_level(tmp_path, compression, ptype, tile):
    rng = np.random.default_rng(compression + ptype)
    h, w = 45, 70                                             # ragged edge tiles in both directions
    dt = np.float16 if ptype == 1 else np.float32
    px = (rng.random((h, w, 4)) * 4).astype(dt)               # stored channel order A,B,G,R
    chans = [("A", ptype), ("B", ptype), ("G", ptype), ("R", ptype)]
    p = tmp_path / "tiled.exr"
    p.write_bytes(_exr_tiled(w, h, chans, compression, px, tile[0], tile[1], shuffle=rng))
    got = mid.load_image(p)
    assert got.shape == (h, w, 4)
    assert np.array_equal(got, px[..., ::-1].astype(np.float32))


def _pxr24_block(lines, channels, w):
    """ImfPxr24Compressor restated independently: per line and channel, delta-coded values split into byte planes
    (MSB plane first): HALF 2 planes, FLOAT 3 planes of the float24 = top 24 bits (rounded), UINT 4 planes; then deflate."""
    out = bytearray()
    for ln in lines:
        for name, t in channels:
            if t == 1:
                v = np.frombuffer(ln[name], np.uint16).astype(np.uint32)
                d = (v - np.concatenate([[0], v[:-1]])).astype(np.uint32)
                out += ((d >> 8) & 255).astype(np.uint8).tobytes() + (d & 255).astype(np.uint8).tobytes()
            elif t == 2:
                v24 = np.frombuffer(ln[name], np.uint32) >> 8                      # caller passes floats whose low 8 bits are zero
                d = (v24 - np.concatenate([[0], v24[:-1]])).astype(np.uint32)
                out += ((d >> 16) & 255).astype(np.uint8).tobytes() + ((d >> 8) & 255).astype(np.uint8).tobytes() + (d & 255).astype(np.uint8).tobytes()
            else:
                v = np.frombuffer(ln[name], np.uint32)
                d = (v - np.concatenate([[0], v[:-1]])).astype(np.uint32)
                out += b"".join(((d >> sh) & 255).astype(np.uint8).tobytes() for sh in (24, 16, 8, 0))
    return zlib.compress(bytes(out))


def test_exr_pxr24_reader(tmp_path):
    """PXR24 scanline file with HALF, FLOAT and UINT channels (FLOAT values pre-truncated to 24 bits, so the lossy codec
    is exact on them); 16-line blocks, ragged last block, non-zero data window origin."""
    rng = np.random.default_rng(24)
    h, w = 37, 53
    f = (rng.standard_normal((h, w, 3)) * 3).astype(np.float32)
    f = (f.view(np.uint32) & np.uint32(0xffffff00)).view(np.float32)               # representable in float24
    a = rng.random((h, w)).astype(np.float16)
    chans = [("A", 1), ("B", 2), ("G", 2), ("R", 2)]
    lines = [{"A": a[y].tobytes(), "B": f[y, :, 2].tobytes(), "G": f[y, :, 1].tobytes(), "R": f[y, :, 0].tobytes()} for y in range(h)]
    blocks = [(y0 - 2, _pxr24_block(lines[y0:y0 + 16], chans, w)) for y0 in range(0, h, 16)]
    blob = bytearray(_exr(w, h, chans, 0, lines, None, data_window=(3, -2, 3 + w - 1, -2 + h - 1)))
    i = blob.index(b"compression\0compression\0") + 24 + 4
    blob[i] = 5
    hdr_end = blob.index(b"screenWindowWidth\0float\0") + 24 + 4 + 4 + 1
    pos = hdr_end + 8 * len(blocks)
    body, offs = b"", []
    for y, payload in blocks:
        offs.append(pos + len(body))
        body += struct.pack("<ii", y, len(payload)) + payload
    p = tmp_path / "pxr24.exr"
    p.write_bytes(bytes(blob[:hdr_end]) + b"".join(struct.pack("<Q", o) for o in offs) + body)
    got = mid.load_image(p)
    assert got.shape == (h, w, 4)
    assert np.array_equal(got[..., :3], f) and np.array_equal(got[..., 3], a.astype(np.float32))
    # UINT channel + corruption
    u = rng.integers(0, 2 ** 32 - 1, (5, 9), dtype=np.uint64).astype(np.uint32)
    ul = [{"Y": u[y].tobytes()} for y in range(5)]
    blob = bytearray(_exr(9, 5, [("Y", 0)], 0, ul, None))
    blob[blob.index(b"compression\0compression\0") + 24 + 4] = 5
    hdr_end = blob.index(b"screenWindowWidth\0float\0") + 24 + 4 + 4 + 1
    payload = _pxr24_block(ul, [("Y", 0)], 9)
    q = tmp_path / "pxr24u.exr"
    q.write_bytes(bytes(blob[:hdr_end]) + struct.pack("<Q", hdr_end + 8) + struct.pack("<ii", 0, len(payload)) + payload)
    assert np.array_equal(mid.load_image(q)[..., 0], u.astype(np.float32))
    bad = bytearray(q.read_bytes())
    bad[-3] ^= 0x55
    q.write_bytes(bytes(bad))
    with pytest.raises(mid.MidError):
        mid.load_image(q)


def test_truncated_and_corrupt_files_are_errors_not_crashes(tmp_path):
    """Every prefix of a valid file, and single-byte corruptions of its header area, must come back
    as MID_ERR_IO (or decode to the right shape) -- never crash or hang."""
    rng = np.random.default_rng(11)
    a8 = rng.integers(0, 256, (9, 11, 4), dtype=np.uint8)
    af = rng.random((18, 11, 4)).astype(np.float32)
    for name, arr in (("t.png", a8), ("t.exr", af)):
        p = tmp_path / name
        mid.save_image(p, arr)
        blob = p.read_bytes()
        q = tmp_path / ("cut_" + name)
        for n in list(range(0, min(len(blob), 400))) + list(range(400, len(blob), 37)):
            q.write_bytes(blob[:n])
            with pytest.raises(mid.MidError) as e:
                mid.load_image(q)
            assert e.value.code == 5
        for i in range(8, min(len(blob), 330)):
            b = bytearray(blob)
            b[i] ^= 0x5a
            q.write_bytes(bytes(b))
            try:
                out = mid.load_image(q)
                assert out.ndim == 3 and out.shape[2] == 4
            except mid.MidError as e:
                assert e.code == 5


def test_sanitizer_sweep(tmp_path):
    """ASan + UBSan build of the decoders (CPU build only) over truncated and corrupted files."""
    import os
    import subprocess
    from conftest import ROOT
    codec = os.path.join(ROOT, "image_denoising_filter_amd", "csrc", "codec")
    exe = tmp_path / "codec_sanitize"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    os.path.join(ROOT, "tools", "codec_sanitize.cpp"), os.path.join(codec, "png.cpp"), os.path.join(codec, "exr.cpp"),
                    os.path.join(codec, "piz.cpp"), "-lz", "-o", str(exe)], check=True)
    # two PIZ files from the test encoder (14-bit and 16-bit wavelet modes) join the sweep
    extra = []
    rng = np.random.default_rng(0)
    for name, px in (("s14.exr", (np.round(rng.random((40, 21, 4)) * 16) / 8).astype(np.float16)),
                     ("s16.exr", (np.round(rng.random((35, 17, 4)) * 4000) * 8).astype(np.float32))):
        chans = [("A", 1 if px.dtype == np.float16 else 2), ("B", 1 if px.dtype == np.float16 else 2),
                 ("G", 1 if px.dtype == np.float16 else 2), ("R", 1 if px.dtype == np.float16 else 2)]
        idx = {"R": 0, "G": 1, "B": 2, "A": 3}
        lines = [{n: np.ascontiguousarray(px[y, :, idx[n]]).tobytes() for n, _ in chans} for y in range(px.shape[0])]
        f = tmp_path / name
        f.write_bytes(_exr(px.shape[1], px.shape[0], chans, 4, lines, None))
        extra.append(str(f))
    # a ZIP tiled file and a PXR24 file (truncations and byte flips of both go through the sanitizer build too)
    tpx = (rng.random((21, 30, 4)) * 2).astype(np.float32)
    ft = tmp_path / "s_tiled.exr"
    ft.write_bytes(_exr_tiled(30, 21, [("A", 2), ("B", 2), ("G", 2), ("R", 2)], 3, tpx, 16, 8))
    extra.append(str(ft))
    hp = (rng.random((20, 11)) * 3).astype(np.float16)
    pl = [{"Y": hp[y].tobytes()} for y in range(20)]
    blob = bytearray(_exr(11, 20, [("Y", 1)], 0, pl, None))
    blob[blob.index(b"compression\0compression\0") + 24 + 4] = 5
    hdr_end = blob.index(b"screenWindowWidth\0float\0") + 24 + 4 + 4 + 1
    pay = [_pxr24_block(pl[y0:y0 + 16], [("Y", 1)], 11) for y0 in (0, 16)]
    o0 = hdr_end + 16
    fp = tmp_path / "s_pxr24.exr"
    fp.write_bytes(bytes(blob[:hdr_end]) + struct.pack("<QQ", o0, o0 + 8 + len(pay[0])) +
                   struct.pack("<ii", 0, len(pay[0])) + pay[0] + struct.pack("<ii", 16, len(pay[1])) + pay[1])
    assert np.array_equal(mid.load_image(fp)[..., 0], hp.astype(np.float32))
    extra.append(str(fp))
    r = subprocess.run([str(exe)] + extra, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitizer sweep done" in r.stdout


# ---- PIZ (decoder checked against the independent test encoder tests/piz_encoder.py) --------------
@pytest.mark.parametrize("ptype", [1, 2])
@pytest.mark.parametrize("shape,kind", [((37, 21), "smooth"), ((64, 33), "smooth"), ((5, 70), "noise"), ((33, 1), "smooth"),
                                        ((40, 48), "flat"), ((35, 19), "wide")])
def test_exr_piz_reader(tmp_path, ptype, shape, kind):
    """HALF and FLOAT channels, odd sizes, several 32-line blocks; 'smooth'/'flat' keep the value LUT below
    2^14 (14-bit wavelet) and produce long runs (run-length symbol), 'wide'/'noise' force the 16-bit wavelet
    and long Huffman codes."""
    rng = np.random.default_rng(shape[0] * 7 + shape[1] + ptype)
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "smooth":
        px = np.stack([np.round(xx * 0.25) / 4, np.round(yy * 0.5) / 8, np.round((xx + yy) * 0.125) / 2, np.ones((h, w))], -1)
    elif kind == "flat":
        px = np.tile(np.float64([0.5, 0.25, 0.125, 1.0]), (h, w, 1))
    elif kind == "wide":
        px = rng.random((h, w, 4)) * 60000 - 30000
    else:
        px = rng.standard_normal((h, w, 4)) * 3
    px = px.astype(np.float16 if ptype == 1 else np.float32)
    chans = [("A", ptype), ("B", ptype), ("G", ptype), ("R", ptype)]
    idx = {"R": 0, "G": 1, "B": 2, "A": 3}
    lines = [{n: np.ascontiguousarray(px[y, :, idx[n]]).tobytes() for n, _ in chans} for y in range(h)]
    for runs in (None, "noruns"):
        p = tmp_path / f"piz_{runs}.exr"
        p.write_bytes(_exr(w, h, chans, 4, lines, runs))
        got = mid.load_image(p)
        assert np.array_equal(got.view(np.uint32), px.astype(np.float32).view(np.uint32)), (kind, runs)


def test_exr_piz_mixed_channel_types_and_corruption(tmp_path):
    rng = np.random.default_rng(3)
    h, w = 45, 23
    r = (np.round(rng.random((h, w)) * 16) / 8).astype(np.float16)       # few distinct values: PIZ really compresses
    g = (np.round(rng.random((h, w)) * 16) / 8).astype(np.float32)
    b = np.round(rng.random((h, w)) * 8).astype(np.float16)
    chans = [("B", 1), ("G", 2), ("R", 1)]
    lines = [{"R": r[y].tobytes(), "G": g[y].tobytes(), "B": b[y].tobytes()} for y in range(h)]
    p = tmp_path / "mixed.exr"
    blob = _exr(w, h, chans, 4, lines, None)
    p.write_bytes(blob)
    got = mid.load_image(p)
    assert np.array_equal(got[..., 0], r.astype(np.float32)) and np.array_equal(got[..., 1], g)
    assert np.array_equal(got[..., 2], b.astype(np.float32)) and np.all(got[..., 3] == 1)
    q = tmp_path / "bad.exr"
    ok = bad = 0
    for i in range(len(blob) - 600, len(blob), 7):            # corrupt the compressed payload: error or same shape, never a crash
        bb = bytearray(blob)
        bb[i] ^= 0xa5
        q.write_bytes(bytes(bb))
        try:
            out = mid.load_image(q)
            assert out.shape == (h, w, 4)
            ok += 1
        except mid.MidError as e:
            assert e.code == 5
            bad += 1
    assert bad > 0


def test_image_threads_is_per_calling_thread_and_does_not_change_the_bytes(tmp_path):
    """mid_image_threads(n): how many host threads an image call of the CALLING thread may use (0 = default, < 0 = query).
    The setting is thread-local, and the files (PNG: 1 MiB deflate segments; EXR: 16-line ZIP chunks) and decoded pixels are the
    same bytes whatever it is -- what mi_denoise --animation relies on when it decodes / encodes one file per worker thread."""
    import threading
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (300, 1200, 4), dtype=np.uint8)          # 1.44 MB raw: two deflate segments
    f32 = (rng.random((70, 90, 4), dtype=np.float32) * 4).astype(np.float32)
    assert mid.lib.mid_image_threads(-1) == 0                           # the default
    files = {}
    for cap in (0, 1, 3):
        assert mid.lib.mid_image_threads(cap) in (0, 1, 3)
        assert mid.lib.mid_image_threads(-1) == cap
        for name, arr in (("a.png", u8), ("a.exr", f32)):
            path = tmp_path / f"{cap}_{name}"
            mid.save_image(path, arr)
            assert np.array_equal(mid.load_image(path), arr)
            files.setdefault(name, []).append(path.read_bytes())
    assert all(b == files["a.png"][0] for b in files["a.png"]) and all(b == files["a.exr"][0] for b in files["a.exr"])
    seen = {}

    def other():
        seen["before"] = mid.lib.mid_image_threads(-1)                  # this thread never set anything: the default
        mid.lib.mid_image_threads(7)
        seen["after"] = mid.lib.mid_image_threads(-1)
    t = threading.Thread(target=other)
    t.start(); t.join()
    assert seen == {"before": 0, "after": 7} and mid.lib.mid_image_threads(-1) == 3
    assert mid.lib.mid_image_threads(0) == 3 and mid.lib.mid_image_threads(-1) == 0


def test_concurrent_codec_calls_under_tsan(tmp_path):
    """ThreadSanitizer build (CPU only) of the codecs called the way mi_denoise --animation calls them: four host threads encode
    and decode different images at once, each with its own mid_image_threads setting (0 = up to 16 inner threads, 1, 2, 3).  No
    data race, every thread gets its own image back, and the file bytes equal the serial encoder's."""
    import os
    import subprocess
    from conftest import ROOT
    codec = os.path.join(ROOT, "image_denoising_filter_amd", "csrc", "codec")
    exe = tmp_path / "codec_tsan"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", os.path.join(ROOT, "tools", "codec_threads_tsan.cpp"),
                    os.path.join(codec, "png.cpp"), os.path.join(codec, "exr.cpp"), os.path.join(codec, "piz.cpp"), "-lz", "-lpthread", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "concurrent codec calls done" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout[-2000:] + r.stderr[-4000:]

