"""Test-only PIZ ENCODER (OpenEXR's LUT + wavelet + Huffman scheme), written independently of
csrc/codec/piz.cpp from the same published description, so that the decoder has something to be
checked against in an environment without any third-party PIZ file.  Pure Python: small images only."""
import heapq
import struct

import numpy as np

HUF_ENCSIZE = (1 << 16) + 1
SHORT_ZEROCODE_RUN, LONG_ZEROCODE_RUN = 59, 63
SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN
LONGEST_LONG_RUN = 255 + SHORTEST_LONG_RUN


class BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.c = 0
        self.lc = 0
        self.nbits = 0

    def put(self, n, bits):
        self.c = (self.c << n) | bits
        self.lc += n
        self.nbits += n
        while self.lc >= 8:
            self.lc -= 8
            self.out.append((self.c >> self.lc) & 0xff)
        self.c &= (1 << self.lc) - 1

    def flush(self):
        if self.lc > 0:
            self.out.append((self.c << (8 - self.lc)) & 0xff)
            self.lc = 0
        return bytes(self.out)


def _wenc14(a, b):
    a = a - 65536 if a >= 32768 else a
    b = b - 65536 if b >= 32768 else b
    m = (a + b) >> 1
    d = a - b
    return m & 0xffff, d & 0xffff


def _wenc16(a, b):
    ao = (a + 32768) & 0xffff
    m = (ao + b) >> 1
    d = ao - b
    if d < 0:
        m = (m + 32768) & 0xffff
    return m & 0xffff, d & 0xffff


def wav2_encode(buf, base, nx, ox, ny, oy, mx):
    """In-place forward wavelet on the flat python list `buf` (one channel component)."""
    enc = _wenc14 if mx < (1 << 14) else _wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py = base
        ey = base + oy * (ny - p2)
        while py <= ey:
            px = py
            ex = py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i01 = enc(buf[px], buf[p01])
                i10, i11 = enc(buf[p10], buf[p11])
                buf[px], buf[p10] = enc(i00, i10)
                buf[p01], buf[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                buf[px], buf[p10] = enc(buf[px], buf[p10])
            py += oy2
        if ny & p:
            px = py
            ex = py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                buf[px], buf[p01] = enc(buf[px], buf[p01])
                px += ox2
        p = p2
        p2 <<= 1


def _code_lengths(freq):
    """Plain Huffman code lengths for the symbols with freq > 0."""
    heap = [(f, i, (s,)) for i, (s, f) in enumerate(sorted(freq.items()))]
    heapq.heapify(heap)
    length = {s: 0 for s in freq}
    if len(heap) == 1:
        length[heap[0][2][0]] = 1
    cnt = len(heap)
    while len(heap) > 1:
        f1, _, s1 = heapq.heappop(heap)
        f2, _, s2 = heapq.heappop(heap)
        for s in s1 + s2:
            length[s] += 1
        cnt += 1
        heapq.heappush(heap, (f1 + f2, cnt, s1 + s2))
    assert max(length.values()) <= 58
    return length


def _canonical(length_by_symbol):
    """length | code << 6 per symbol; codes assigned exactly as the format prescribes."""
    n = [0] * 59
    n[0] = HUF_ENCSIZE - len(length_by_symbol)
    for l in length_by_symbol.values():
        n[l] += 1
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        n[i] = c
        c = nc
    codes = {}
    for s in sorted(length_by_symbol):
        l = length_by_symbol[s]
        if l > 0:
            codes[s] = (l, n[l])
            n[l] += 1
    return codes


def huf_compress(symbols, use_runs=True):
    symbols = [int(s) for s in symbols]
    freq = {}
    for s in symbols:
        freq[s] = freq.get(s, 0) + 1
    im = min(freq)
    rlc = max(freq) + 1                       # pseudo-symbol for run lengths
    freq[rlc] = 1
    iM = rlc
    lengths = _code_lengths(freq)
    codes = _canonical(lengths)
    # pack the table of 6-bit lengths with zero runs
    tw = BitWriter()
    s = im
    while s <= iM:
        l = lengths.get(s, 0)
        if l == 0:
            zerun = 1
            while s + zerun <= iM and zerun < LONGEST_LONG_RUN and lengths.get(s + zerun, 0) == 0:
                zerun += 1
            if zerun >= 2:
                if zerun >= SHORTEST_LONG_RUN:
                    tw.put(6, LONG_ZEROCODE_RUN)
                    tw.put(8, zerun - SHORTEST_LONG_RUN)
                else:
                    tw.put(6, SHORT_ZEROCODE_RUN + zerun - 2)
                s += zerun
                continue
        tw.put(6, l)
        s += 1
    table = tw.flush()
    # encode the data
    dw = BitWriter()

    def code(sym):
        l, c = codes[sym]
        dw.put(l, c)
    i = 0
    while i < len(symbols):
        s = symbols[i]
        run = 0
        while i + 1 + run < len(symbols) and symbols[i + 1 + run] == s and run < 255:
            run += 1
        ls, lr = codes[s][0], codes[rlc][0]
        if use_runs and run > 0 and ls + lr + 8 < ls * run:
            code(s)
            code(rlc)
            dw.put(8, run)
        else:
            for _ in range(run + 1):
                code(s)
        i += run + 1
    nbits = dw.nbits
    data = dw.flush()
    return struct.pack("<IIIII", im, iM, len(table), nbits, 0) + table + data


def piz_block(lines, channels, width, use_runs=True):
    """lines[y][name] = bytes of one scanline of one channel; channels = [(name, type)] in file order.
    Returns the compressed chunk payload (or the raw scanline-interleaved bytes when that is smaller)."""
    nl = len(lines)
    raw = b"".join(lines[y][n] for y in range(nl) for n, _ in channels)
    # channel-major 16-bit words
    words = []
    starts, sizes = [], []
    for n, t in channels:
        size = 1 if t == 1 else 2
        starts.append(len(words))
        sizes.append(size)
        for y in range(nl):
            words.extend(np.frombuffer(lines[y][n], dtype="<u2").tolist())
    bitmap = bytearray(8192)
    for v in words:
        bitmap[v >> 3] |= 1 << (v & 7)
    bitmap[0] &= 0xfe
    nz = [i for i, b in enumerate(bitmap) if b]
    min_nz, max_nz = (nz[0], nz[-1]) if nz else (8191, 0)
    lut, k = [0] * 65536, 0
    for i in range(65536):
        if i == 0 or (bitmap[i >> 3] >> (i & 7)) & 1:
            lut[i] = k
            k += 1
    max_value = k - 1
    words = [lut[v] for v in words]
    for st, size in zip(starts, sizes):
        for j in range(size):
            wav2_encode(words, st + j, width, size, nl, width * size, max_value)
    huf = huf_compress(words, use_runs)
    out = struct.pack("<HH", min_nz, max_nz)
    if min_nz <= max_nz:
        out += bytes(bitmap[min_nz:max_nz + 1])
    out += struct.pack("<i", len(huf)) + huf
    return out if len(out) < len(raw) else raw
