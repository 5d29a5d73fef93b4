"""Seeded synthetic DEFLATE workloads shared by bench.py and the tests.

Nothing here is read from the reference tree at run time.  The generators restate the
benchmark inputs of the build's measurement plan:

* text streams   -- English-like text over a fixed vocabulary of 4096 pseudo-words drawn
                    with a Zipf law, compressed with zlib level 6 (dynamic Huffman blocks);
* stored streams -- incompressible payload wrapped in stored blocks (65535 + rest) and a
                    zlib wrapper, the layout the reference's fixture generator uses for its
                    `uncompressed_*` files (test/data/gendata.py:27-56);
* fixed streams  -- the text compressed with Z_FIXED (fixed Huffman blocks only);
* PNG-like streams -- filtered RGBA scanlines, zlib level 6, cut into IDAT-sized pieces.
"""
import struct
import zlib

import numpy as np

MASK64 = (1 << 64) - 1
SEED_BASE = 0xDEF1A7E


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def stream_seed(i: int) -> int:
    return splitmix64(SEED_BASE ^ i)


class TextGen:
    """4096 pseudo-words (2..12 letters), Zipf(s=1) ranks, separators ' ', ', ', '. ', '\\n'."""

    def __init__(self, vocab=4096, s=1.0):
        rng = np.random.default_rng(splitmix64(SEED_BASE))
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        lw = 1.0 / np.arange(1, 27)
        lw /= lw.sum()
        self.words = []
        for _ in range(vocab):
            n = int(rng.integers(2, 13))
            self.words.append(bytes(rng.choice(letters, size=n, p=lw)))
        p = 1.0 / np.arange(1, vocab + 1) ** s
        self.cdf = np.cumsum(p / p.sum())
        self.seps = [b" ", b", ", b". ", b"\n"]
        self.sep_cdf = np.cumsum([0.82, 0.08, 0.07, 0.03])
        self._table = None

    def make_joined(self, i: int, size: int) -> bytes:
        """The definition: words and separators joined one after the other (slow: a Python loop over the words)."""
        rng = np.random.default_rng(stream_seed(i))
        nwords = size // 3 + 8
        wi = np.searchsorted(self.cdf, rng.random(nwords))
        si = np.searchsorted(self.sep_cdf, rng.random(nwords))
        words, seps = self.words, self.seps
        out = b"".join([words[min(a, len(words) - 1)] + seps[min(b, 3)] for a, b in zip(wi.tolist(), si.tolist())])
        while len(out) < size:
            out += out
        return out[:size]

    def _token_table(self):
        """All words and separators back to back, with every token's start and length (tokens 0 .. V-1 are the
        words, V .. V+3 the separators)."""
        toks = self.words + self.seps
        lens = np.array([len(t) for t in toks], dtype=np.int64)
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        return np.frombuffer(b"".join(toks), dtype=np.uint8), starts, lens

    def make(self, i: int, size: int) -> bytes:
        """The same bytes as make_joined(), gathered with array operations (the same random draws in the same order;
        the benchmark builds thousands of these streams per run)."""
        if self._table is None:
            self._table = self._token_table()
        blob, tstart, tlen = self._table
        v = len(self.words)
        rng = np.random.default_rng(stream_seed(i))
        nwords = size // 3 + 8
        wi = np.minimum(np.searchsorted(self.cdf, rng.random(nwords)), v - 1)
        si = np.minimum(np.searchsorted(self.sep_cdf, rng.random(nwords)), 3)
        tok = np.empty(2 * nwords, dtype=np.int64)
        tok[0::2] = wi
        tok[1::2] = v + si
        lens = tlen[tok]
        ends = np.cumsum(lens)
        total = int(ends[-1])
        k = min(len(tok), int(np.searchsorted(ends, size)) + 1)         # tokens that reach the first `size` bytes
        begin = ends[:k] - lens[:k]
        idx = np.repeat(tstart[tok[:k]] - begin, lens[:k]) + np.arange(int(ends[k - 1]), dtype=np.int64)
        out = blob[idx]
        if total < size:                                                # (the joined text doubled until it is long enough)
            out = np.resize(out, size)
        return out[:size].tobytes()


_textgen = None


def text_stream(i: int, size: int = 65536) -> bytes:
    global _textgen
    if _textgen is None:
        _textgen = TextGen()
    return _textgen.make(i, size)


def deflate(raw: bytes, level=6, zlib_wrap=True, strategy=zlib.Z_DEFAULT_STRATEGY) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, 15 if zlib_wrap else -15, 8, strategy)
    return c.compress(raw) + c.flush()


def stored_stream(i: int, size: int = 65536, zlib_wrap=True) -> tuple:
    """(raw, compressed): random payload as stored blocks of at most 65535 bytes."""
    rng = np.random.default_rng(stream_seed(i) ^ 0x5705ED)
    raw = rng.integers(0, 256, size=size, dtype=np.uint8).tobytes()
    out = bytearray(b"\x78\x01" if zlib_wrap else b"")
    pos = 0
    while True:
        n = min(65535, size - pos)
        final = pos + n >= size
        out += struct.pack("<BHH", 1 if final else 0, n, n ^ 0xFFFF) + raw[pos:pos + n]
        pos += n
        if final:
            break
    if zlib_wrap:
        out += struct.pack(">I", zlib.adler32(raw))
    return raw, bytes(out)


def png_like(i: int, width=128, height=128) -> bytes:
    """RGBA scanlines with a leading filter byte per row: gradient plus seeded low-bit noise."""
    rng = np.random.default_rng(stream_seed(i) ^ 0x1DA7)
    y, x = np.mgrid[0:height, 0:width]
    img = np.empty((height, width, 4), dtype=np.uint8)
    img[..., 0] = (x * 255 // max(1, width - 1)).astype(np.uint8)
    img[..., 1] = (y * 255 // max(1, height - 1)).astype(np.uint8)
    img[..., 2] = ((x + y) * 255 // max(1, width + height - 2)).astype(np.uint8)
    img[..., 3] = 255
    img[..., :3] ^= rng.integers(0, 4, size=(height, width, 3), dtype=np.uint8)
    rows = np.zeros((height, 1 + width * 4), dtype=np.uint8)
    rows[:, 1:] = img.reshape(height, width * 4)
    return rows.tobytes()


def png_filter(pixels: np.ndarray, bpp: int, types) -> bytes:
    """The encoder's side of PNG filtering (PNG specification, section 9): pixels = (rows, rowbytes) uint8,
    types[r] = filter type of scanline r (0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth).  -> rows of 1 + rowbytes bytes."""
    rows, rowbytes = pixels.shape
    cur = pixels.astype(np.int32)
    a = np.zeros_like(cur)
    a[:, bpp:] = cur[:, :rowbytes - bpp] if rowbytes > bpp else 0
    b = np.zeros_like(cur)
    b[1:] = cur[:-1]
    c = np.zeros_like(cur)
    c[1:, bpp:] = cur[:-1, :rowbytes - bpp] if rowbytes > bpp else 0
    p = a + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
    pred = [np.zeros_like(cur), a, b, (a + b) >> 1, paeth]
    out = np.zeros((rows, 1 + rowbytes), dtype=np.uint8)
    for r in range(rows):
        t = int(types[r])
        out[r, 0] = t
        out[r, 1:] = ((cur[r] - pred[t][r]) & 0xFF).astype(np.uint8)
    return out.tobytes()


def png_filtered(i: int, width=128, height=128, bpp=4, types=None):
    """png_like()'s picture at bpp bytes per pixel with real filter types: by default the five types in a
    seeded shuffle, so that every type meets every type above it.  -> (pixel bytes, filtered bytes, types)"""
    rng = np.random.default_rng(stream_seed(i) ^ 0xF117E4)
    y, x = np.mgrid[0:height, 0:width]
    img = np.empty((height, width, bpp), dtype=np.uint8)
    for ch in range(bpp):
        img[..., ch] = ((x * (ch + 1) + y * (bpp - ch)) * 255 // max(1, (width + height) * bpp)).astype(np.uint8)
    img ^= rng.integers(0, 8, size=img.shape, dtype=np.uint8)
    if height and width:
        img[rng.integers(0, height), :, :] = rng.integers(0, 256, size=(width, bpp), dtype=np.uint8)   # one noisy line
    pixels = img.reshape(height, width * bpp)
    if types is None:
        types = rng.integers(0, 5, size=height)
    return pixels.tobytes(), png_filter(pixels, bpp, types), [int(t) for t in types]


def idat_pieces(i: int, comp: bytes) -> list:
    """Cut a zlib stream into IDAT-like pieces: 50 % exactly 8192 B, 25 % 1..64 B, 25 % 64..4096 B."""
    rng = np.random.default_rng(stream_seed(i) ^ 0x1DA7C)
    out, pos = [], 0
    while pos < len(comp):
        r = rng.random()
        n = 8192 if r < 0.5 else int(rng.integers(1, 65)) if r < 0.75 else int(rng.integers(64, 4097))
        out.append(comp[pos:pos + n])
        pos += n
    return out


def pack(streams, caps):
    """List of compressed byte strings + output capacities -> (src arena, src_off, dst_off)."""
    n = len(streams)
    src_off = np.zeros(n + 1, dtype=np.uint64)
    dst_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in streams], out=src_off[1:])
    np.cumsum(list(caps), out=dst_off[1:])
    src = np.frombuffer(b"".join(streams), dtype=np.uint8) if n and src_off[-1] else np.zeros(1, np.uint8)
    return src, src_off, dst_off
