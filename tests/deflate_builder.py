"""Bit-level writer for hand-built DEFLATE dynamic blocks (test input generator).

Used to reach header and table-build branches that a real compressor never emits:
over-subscribed / incomplete code-length sets, repeat codes at position 0 or past the end,
symbols that have a code but no table entry (286, 287, distance 30, 31), a distance code
set that overflows the reference's 512-entry distance table, distances before the start
of the output.
"""

ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]

# complete code-length code: symbols 0..15 -> 5 bits, 16 -> 3, 17 -> 3, 18 -> 2
CL_DEFAULT = [5] * 16 + [3, 3, 2]

# distance length set (Kraft sum 0.90) whose subtables need more than 512 entries
DIST_OVERFLOW = [15, 15, 15, 10, 8, 2, 15, 15, 15, 13, 12, 9, 6, 3, 12, 15, 15, 9, 5, 6, 7, 9, 13, 4, 7,
                 2, 15, 15, 13, 3]


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, value, count):          # LSB first
        self.acc |= (value & ((1 << count) - 1)) << self.n
        self.n += count
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def code(self, code, length):          # Huffman codes go MSB first
        for i in range(length - 1, -1, -1):
            self.bits((code >> i) & 1, 1)

    def done(self):
        if self.n:
            self.out.append(self.acc & 0xFF)
            self.acc, self.n = 0, 0
        return bytes(self.out)


def canonical(lens):
    """code per symbol (None where length 0); works for incomplete sets, garbage if over-subscribed"""
    count = [0] * 17
    for l in lens:
        count[l] += 1
    count[0] = 0
    nxt, code = [0] * 17, 0
    for l in range(1, 16):
        code = (code + count[l - 1]) << 1
        nxt[l] = code
    out = []
    for l in lens:
        if l == 0:
            out.append(None)
        else:
            out.append(nxt[l] & ((1 << l) - 1))
            nxt[l] += 1
    return out


def dynamic_block(lit_lens, dist_lens, symbols, cl_lens=None, cl_seq=None, final=True, hclen=19,
                  eob=True):
    """One dynamic block.

    symbols: list of ('lit', byte) | ('sym', litlen symbol, extra value, extra bits)
             | ('dist', dist symbol, extra value, extra bits)
    cl_seq:  explicit code-length-alphabet sequence [(sym, extra value)] replacing the plain
             one-symbol-per-length encoding of lit_lens + dist_lens.
    """
    cl_lens = list(cl_lens or CL_DEFAULT)
    w = BitWriter()
    w.bits(1 if final else 0, 1)
    w.bits(2, 2)
    w.bits(len(lit_lens) - 257, 5)
    w.bits(len(dist_lens) - 1, 5)
    w.bits(hclen - 4, 4)
    for i in range(hclen):
        w.bits(cl_lens[ORDER[i]], 3)
    cl_codes = canonical(cl_lens)
    if cl_seq is None:
        cl_seq = [(l, 0) for l in list(lit_lens) + list(dist_lens)]
    for sym, extra in cl_seq:
        w.code(cl_codes[sym] or 0, cl_lens[sym])
        if sym == 16:
            w.bits(extra, 2)
        elif sym == 17:
            w.bits(extra, 3)
        elif sym == 18:
            w.bits(extra, 7)
    lc, dc = canonical(lit_lens), canonical(dist_lens)
    for s in symbols:
        if s[0] == "lit":
            w.code(lc[s[1]], lit_lens[s[1]])
        elif s[0] == "sym":
            w.code(lc[s[1]], lit_lens[s[1]])
            w.bits(s[2], s[3])
        else:
            w.code(dc[s[1]], dist_lens[s[1]])
            w.bits(s[2], s[3])
    if eob and lit_lens[256]:
        w.code(lc[256], lit_lens[256])
    return w.done()


def _small_alphabet():
    """'a' 'b' 'c' 2 bits each... : a,b -> 2 bits, c, EOB, 257(len 3), 285(len 258) -> 3 bits; complete"""
    lit = [0] * 286
    lit[ord("a")] = 2
    lit[ord("b")] = 2
    lit[ord("c")] = 3
    lit[256] = 3
    lit[257] = 3
    lit[285] = 3
    dist = [1, 1]                          # distance 1 and 2, complete
    return lit, dist


def header_cases():
    cases = []
    lit, dist = _small_alphabet()
    abab = [("lit", 97), ("lit", 98), ("sym", 257, 0, 0), ("dist", 1, 0, 0), ("lit", 99),
            ("sym", 285, 0, 0), ("dist", 0, 0, 0)]
    valid = dynamic_block(lit, dist, abab)
    cases.append(valid)
    cases += [valid[:k] for k in range(len(valid))]                      # every truncation
    cases.append(dynamic_block(lit, dist, abab, final=False) + valid)    # two blocks
    # distance before the start of the output
    cases.append(dynamic_block(lit, dist, [("lit", 97), ("sym", 257, 0, 0), ("dist", 1, 0, 0)]))
    cases.append(dynamic_block(lit, dist, [("sym", 257, 0, 0), ("dist", 0, 0, 0)]))
    # code-length code: over-subscribed, single code, two codes (incomplete)
    cases.append(dynamic_block(lit, dist, abab, cl_lens=[1] * 19))
    cases.append(dynamic_block(lit, dist, abab, cl_lens=[7] * 19))
    one = [0] * 19
    one[8] = 1
    cases.append(dynamic_block([8] * 257, [8], [("lit", 1)], cl_lens=one))
    two = [0] * 19
    two[8], two[0] = 1, 2
    l8 = [8] * 255 + [0, 8]
    cases.append(dynamic_block(l8, [0], [("lit", 65), ("lit", 66), ("lit", 254)], cl_lens=two))
    cases.append(dynamic_block(l8, [0], [("lit", 65)], cl_lens=two, hclen=4))   # most lengths unread
    # all-zero code lengths everywhere
    cases.append(dynamic_block([0] * 257, [0], [], cl_lens=[0] * 19))
    cases.append(dynamic_block([0] * 257, [0], []))
    # repeat codes: 16 at position 0, 16/17/18 running past the end, legal long runs
    n = 257 + 1
    cases.append(dynamic_block([0] * 257, [0], [], cl_seq=[(16, 0)] + [(0, 0)] * n))
    cases.append(dynamic_block([0] * 257, [0], [], cl_seq=[(8, 0)] + [(16, 3)] * 43))
    cases.append(dynamic_block([0] * 257, [0], [], cl_seq=[(18, 127), (18, 127)]))
    cases.append(dynamic_block([0] * 257, [0], [], cl_seq=[(18, 127), (18, 109), (17, 0)]))
    legal = [(18, 127), (18, 107)] + [(2, 0), (16, 0)] + [(17, 7)]      # 138+118 zeros, 256..259 = 2, 10 zeros?
    cases.append(dynamic_block([0] * 260 + [0] * 9, [0], [], cl_seq=legal))
    run_lit = [0] * 256 + [1, 1] + [0] * 12
    cases.append(dynamic_block(run_lit, [0], [("sym", 257, 0, 0)],
                               cl_seq=[(18, 127), (18, 107), (1, 0), (1, 0), (17, 7), (17, 0)]))
    # over-subscribed literal and distance sets
    bad = list(lit)
    bad[ord("d")] = 1
    bad[ord("e")] = 1
    cases.append(dynamic_block(bad, dist, [("lit", 97)]))
    cases.append(dynamic_block(lit, [1, 1, 1], abab))
    # incomplete literal set: unassigned pattern never hit / hit
    inc = [0] * 257
    inc[ord("a")] = 2
    inc[256] = 2
    ok_inc = dynamic_block(inc, [0], [("lit", 97), ("lit", 97)])
    cases.append(ok_inc)
    hdr = dynamic_block(inc, [0], [], eob=False)
    cases.append(hdr[:-1] + bytes([hdr[-1] | 0xC0]) + b"\xff\xff")       # runs into code 11 = unassigned
    # symbols with a code but no table entry: 286 and distance 30
    l288 = [0] * 288
    l288[ord("a")] = 2
    l288[256] = 2
    l288[257] = 2
    l288[286] = 2
    d32 = [0] * 32
    d32[0] = 1
    d32[30] = 1
    cases.append(dynamic_block(l288, d32, [("lit", 97), ("sym", 286, 0, 0)]))
    cases.append(dynamic_block(l288, d32, [("lit", 97), ("sym", 257, 0, 0), ("dist", 30, 0, 0)]))
    cases.append(dynamic_block(l288, d32, [("lit", 97), ("sym", 257, 0, 0), ("dist", 0, 0, 0)]))
    # distance table capacity overflow (rejected while the table is built)
    cases.append(dynamic_block(lit, DIST_OVERFLOW, abab[:2]))
    # missing end-of-block code: block can never finish
    noeob = list(lit)
    noeob[256] = 0
    noeob[ord("d")] = 3
    cases.append(dynamic_block(noeob, dist, abab))
    # long codes (subtables on both tables) with every length 1..15
    deep = [0] * 286
    for k, s in enumerate(range(40, 54)):
        deep[s] = k + 1
    deep[256] = 15
    deep[54] = 15                          # two 15-bit codes complete the set
    ddeep = [0] * 30
    for k in range(14):
        ddeep[k] = k + 1
    ddeep[14] = 15
    ddeep[15] = 15
    deep[257] = 0
    syms = [("lit", s) for s in range(40, 55)] * 3
    cases.append(dynamic_block(deep, ddeep, syms))
    deep2 = list(deep)
    deep2[54] = 0
    deep2[258] = 15
    syms2 = [("lit", s) for s in range(40, 54)] * 20
    for dsym in range(16):
        xb = 0 if dsym < 4 else (dsym >> 1) - 1
        syms2 += [("sym", 258, 0, 0), ("dist", dsym, (1 << xb) - 1 if xb else 0, xb)]
    cases.append(dynamic_block(deep2, ddeep, syms2))
    # "at least 7 bits in the reservoir before every code-length symbol" (infl.c:947), whatever the
    # symbol's own length: the input ends 6 bits (rejected) or more (accepted) before the last one
    short7 = bytes.fromhex("05c0810c00000000307feb06")
    cases += [short7, short7 + b"\x00", short7[:-1], short7 + b"\xff\xff"]
    return cases
