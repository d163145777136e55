"""A small FLAC *encoder* for the tests: writes streams that exercise the decoder paths LibriSpeech's own
files (mono, 16 bit, LPC / FIXED + Rice) never reach — every subframe type, both Rice flavours, escape
partitions, wasted bits, the four channel assignments, explicit block-size / sample-rate fields,
multi-byte frame numbers, 8 / 16 / 24 bit.  Written from the format description (RFC 9639), independent
of csrc/flac_decode.hip.  Test infrastructure only."""
import hashlib

import numpy as np


class BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value: int, bits: int):
        if bits == 0:
            return
        self.acc = (self.acc << bits) | (int(value) & ((1 << bits) - 1))
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xff)
        self.acc &= (1 << self.n) - 1

    def unary(self, q: int):
        while q >= 32:
            self.put(0, 32)
            q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    n_cont = 1
    while v >= 1 << (6 * n_cont + (6 - n_cont)):
        n_cont += 1
    lead = ((0xff << (7 - n_cont)) & 0xff) | (v >> (6 * n_cont))
    return bytes([lead] + [0x80 | ((v >> (6 * i)) & 0x3f) for i in range(n_cont - 1, -1, -1)])


def _residual(bw: BitWriter, res, block: int, order: int, rice2: bool, porder: int, escape: bool):
    bw.put(1 if rice2 else 0, 2)
    bw.put(porder, 4)
    pbits, esc = (5, 31) if rice2 else (4, 15)
    i = 0
    for part in range(1 << porder):
        count = (block >> porder) - (order if part == 0 else 0)
        chunk = [int(x) for x in res[i:i + count]]
        i += count
        if escape and part % 2 == 0:
            raw = max([1] + [(abs(v) if v >= 0 else abs(v) - 1).bit_length() + 1 for v in chunk])
            bw.put(esc, pbits)
            bw.put(raw, 5)
            for v in chunk:
                bw.put(v, raw)
        else:
            mean = (sum(abs(v) for v in chunk) // max(1, len(chunk))) if chunk else 0
            k = min(max(mean.bit_length(), 0), esc - 1)
            bw.put(k, pbits)
            for v in chunk:
                u = (v << 1) if v >= 0 else ((-v) << 1) - 1
                bw.unary(u >> k)
                bw.put(u & ((1 << k) - 1), k)


def _subframe(bw: BitWriter, s, bps: int, kind: str, wasted: int, rice2: bool, porder: int, escape: bool, lpc=None):
    s = [int(x) for x in s]
    block = len(s)
    code = {"constant": 0, "verbatim": 1}.get(kind)
    order = 0
    if kind.startswith("fixed"):
        order = int(kind[5:])
        code = 8 + order
    elif kind == "lpc":
        order = len(lpc["coefs"])
        code = 32 + order - 1
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
        assert all(v % (1 << wasted) == 0 for v in s)
        s = [v >> wasted for v in s]
        bps -= wasted
    else:
        bw.put(0, 1)
    if kind == "constant":
        assert len(set(s)) == 1
        bw.put(s[0], bps)
        return
    if kind == "verbatim":
        for v in s:
            bw.put(v, bps)
        return
    for v in s[:order]:
        bw.put(v, bps)
    if kind == "lpc":
        coefs, shift, precision = lpc["coefs"], lpc["shift"], lpc["precision"]
        bw.put(precision - 1, 4)
        bw.put(shift, 5)
        for c in coefs:
            bw.put(c, precision)
        res = [s[i] - (sum(c * s[i - 1 - j] for j, c in enumerate(coefs)) >> shift) for i in range(order, block)]
    else:
        binom = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        res = [s[i] - sum(c * s[i - 1 - j] for j, c in enumerate(binom)) for i in range(order, block)]
    _residual(bw, res, block, order, rice2, porder, escape)


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12}
BITS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}


def encode_flac(samples: np.ndarray, bps: int, frames, sample_rate: int = 16000, first_frame_number: int = 0,
                known_total: bool = True, with_md5: bool = True) -> bytes:
    """samples int [n, channels]; frames = list of dicts, one per frame, consumed in order:
       block (samples), kind ('constant'|'verbatim'|'fixedK'|'lpc') or kinds per channel, assignment (0..10),
       wasted, rice2, porder, escape, lpc, explicit_block (use the 8/16-bit block size field), explicit_rate,
       bits_from_streaminfo."""
    samples = np.asarray(samples, dtype=np.int64)
    n, nch = samples.shape
    out = bytearray(b"fLaC")
    width = (bps + 7) // 8
    raw = samples.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width].tobytes()
    md5 = hashlib.md5(raw).digest() if with_md5 else bytes(16)
    blocks = [f["block"] for f in frames]
    assert sum(blocks) == n
    si = BitWriter()
    si.put(min(blocks), 16); si.put(max(blocks), 16); si.put(0, 24); si.put(0, 24)
    si.put(sample_rate, 20); si.put(nch - 1, 3); si.put(bps - 1, 5); si.put(n if known_total else 0, 36)
    info = bytes(si.out) + md5
    # a padding block first, STREAMINFO flagged... STREAMINFO must come first: [STREAMINFO][PADDING last]
    out += bytes([0x00]) + len(info).to_bytes(3, "big") + info
    out += bytes([0x81]) + (6).to_bytes(3, "big") + bytes(6)
    pos = 0
    for fi, f in enumerate(frames):
        block, assign = f["block"], f.get("assignment", nch - 1)
        chunk = samples[pos:pos + block]
        pos += block
        bw = BitWriter()
        bw.put(0x3ffe, 14); bw.put(0, 1); bw.put(0, 1)
        explicit_block = f.get("explicit_block", block not in BLOCK_CODES)
        bcode = (6 if block <= 256 else 7) if explicit_block else BLOCK_CODES[block]
        rcode = 13 if f.get("explicit_rate") else 0
        bw.put(bcode, 4); bw.put(rcode, 4); bw.put(assign, 4)
        bw.put(0 if f.get("bits_from_streaminfo") else BITS_CODES[bps], 3); bw.put(0, 1)
        for b in utf8_number(first_frame_number + fi):
            bw.put(b, 8)
        if explicit_block:
            bw.put(block - 1, 8 if block <= 256 else 16)
        if rcode == 13:
            bw.put(sample_rate, 16)
        bw.put(crc8(bytes(bw.out)), 8)
        if assign < 8:
            chans, widths = [chunk[:, c] for c in range(nch)], [bps] * nch
        else:
            left, right = chunk[:, 0], chunk[:, 1]
            side = left - right
            if assign == 8:
                chans, widths = [left, side], [bps, bps + 1]
            elif assign == 9:
                chans, widths = [side, right], [bps + 1, bps]
            else:
                chans, widths = [(left + right) >> 1, side], [bps, bps + 1]
        kinds = f.get("kinds") or [f.get("kind", "verbatim")] * len(chans)
        for c, (s, w) in enumerate(zip(chans, widths)):
            _subframe(bw, s, w, kinds[c], f.get("wasted", 0), f.get("rice2", False), f.get("porder", 0),
                      f.get("escape", False), f.get("lpc"))
        bw.align()
        bw.put(crc16(bytes(bw.out)), 16)
        out += bw.out
    return bytes(out)
