"""An independent reader of OpenCC's .ocd2 dictionaries (TEST INFRASTRUCTURE): every (key, [values]) of the file.

The product's reader (whisper.axera_amd/csrc/t2s.hpp) looks keys up BACKWARDS — key id -> terminal node -> walk to the root —
and keeps the first value only. This one is written separately, in another language and the other way round: a FORWARD
depth-first walk of the LOUDS bit vector from the root (children of node n start behind the n-th 0 bit), which yields every
key with its id (rank of its terminal bit), and a plain sequential read of the serialized value table (all values, not only
the first). tests/test_t2s.py sends every key through AX_WHISPER_ConvertT2S and compares with values[0] read here.

File layout (OpenCC 1.1 SerializedValues + marisa-trie 0.2.x image): "OPENCC_MARISA_0.2.5", "We love Marisa.\\0", then per trie
level: louds / terminal / link bit vectors (u64-size-prefixed unit vector, u32 size, u32 ones, three index vectors), bases
(bytes), extras (flag vector), tail buffer, tail end flags; then per level (innermost first) cache, u32 root child count, u32
config; then u32 item count, u32 value bytes, the NUL-terminated values, and per key a u16 value count + a u16 byte length each."""
import struct

import numpy as np


class _R:
    def __init__(self, b, p):
        self.b, self.p = b, p

    def u32(self):
        v = struct.unpack_from("<I", self.b, self.p)[0]
        self.p += 4
        return v

    def u16(self):
        v = struct.unpack_from("<H", self.b, self.p)[0]
        self.p += 2
        return v

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.p)[0]
        self.p += 8
        return v

    def vec(self):
        n = self.u64()
        d = self.b[self.p:self.p + n]
        self.p += n + (8 - n % 8) % 8
        return d


class _Bits:
    def __init__(self, r):
        units = r.vec()
        self.size, self.num1 = r.u32(), r.u32()
        r.vec(); r.vec(); r.vec()   # rank / select0 / select1 indices: not needed for a sequential walk
        bits = np.unpackbits(np.frombuffer(units, dtype=np.uint8), bitorder="little")[: self.size]
        self.bits = bits.astype(bool)
        self.ones = np.flatnonzero(self.bits)
        self.zeros = np.flatnonzero(~self.bits)
        self.rank1 = np.concatenate([[0], np.cumsum(self.bits)])  # rank1[i] = ones in [0, i)


class _Flags:
    def __init__(self, r):
        self.units = np.frombuffer(r.vec() + b"\0" * 8, dtype="<u8")
        self.value_size, self.mask = r.u32(), r.u32()
        r.u64()

    def get(self, i):
        pos = i * self.value_size
        w, o = pos >> 6, pos & 63
        v = int(self.units[w]) >> o
        if o + self.value_size > 64:
            v |= int(self.units[w + 1]) << (64 - o)
        return v & self.mask


class _Level:
    def __init__(self, r):
        self.louds, self.terminal, self.link = _Bits(r), _Bits(r), _Bits(r)
        self.bases = r.vec()
        self.extras = _Flags(r)
        self.tail = r.vec()
        self.tail_end = _Bits(r)
        self.next = None
        self.num_l1 = 0

    def link_string(self, node):
        lnk = self.bases[node] | (self.extras.get(int(self.link.rank1[node])) << 8)
        if self.next is not None:
            return self.next.spell_up(lnk)
        out = bytearray()
        o = lnk
        if self.tail_end.size == 0:
            while o < len(self.tail) and self.tail[o]:
                out.append(self.tail[o]); o += 1
        else:
            while o < len(self.tail):
                out.append(self.tail[o])
                if self.tail_end.bits[o]:
                    break
                o += 1
        return bytes(out)

    def spell_up(self, node):   # a node of a next-level trie spells its string towards the root
        out = bytearray()
        while True:
            out += self.link_string(node) if self.link.bits[node] else bytes([self.bases[node]])
            if node <= self.num_l1:
                return bytes(out)
            node = int(self.louds.ones[node]) - node - 1

    def label(self, node):
        return self.link_string(node) if self.link.bits[node] else bytes([self.bases[node]])


def read_ocd2(path):
    """-> list of (key bytes, [value bytes, ...]) in key-id order."""
    b = open(path, "rb").read()
    magic = b"OPENCC_MARISA_0.2.5"
    assert b.startswith(magic) and b[len(magic):len(magic) + 16] == b"We love Marisa.\0", "not an ocd2 file"
    r = _R(b, len(magic) + 16)
    levels = [_Level(r)]
    while levels[-1].link.num1 and not len(levels[-1].tail) and len(levels) < 16:
        levels.append(_Level(r))
        levels[-2].next = levels[-1]
    for lv in reversed(levels):
        r.vec()
        lv.num_l1 = r.u32()
        r.u32()
    top = levels[0]
    n_keys = top.terminal.num1
    # forward depth-first walk: node 0 is the root; the children of node n are the 1 bits behind the n-th 0 bit, child id =
    # position - n - 1
    keys = [None] * n_keys
    stack = [(0, b"")]
    while stack:
        node, prefix = stack.pop()
        if top.terminal.bits[node]:
            keys[int(top.terminal.rank1[node])] = prefix
        pos = int(top.louds.zeros[node]) + 1
        while pos < top.louds.size and top.louds.bits[pos]:
            child = pos - node - 1
            stack.append((child, prefix + top.label(child)))
            pos += 1
    assert all(k is not None for k in keys), "a key id without a terminal node"
    n_items, total = r.u32(), r.u32()
    assert n_items == n_keys, (n_items, n_keys)
    vbuf = r.p
    r.p += total
    out, off = [], 0
    for i in range(n_items):
        vals = []
        for _ in range(r.u16()):
            ln = r.u16()
            vals.append(b[vbuf + off:vbuf + off + ln - 1])
            assert b[vbuf + off + ln - 1] == 0
            off += ln
        out.append((keys[i], vals))
    assert off == total and r.p == len(b), (off, total, r.p, len(b))
    return out
