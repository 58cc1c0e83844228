"""A SECOND restatement of the reference's string index, in plain Python, written from index.cpp / index.h alone and
sharing nothing with oracle/cpu_ref.cpp.  Test infrastructure: two independent restatements that agree entry for entry
on radix nodes, leaves, the signed symbol order and the query's probe sequence pin the oracle where the reference
itself cannot be run (its sources need <format>, absent from this image).  Small inputs only.

Citations are /root/reference/src paths:
  * entry = (offset << bits) | doc, doc-major fill, masks grown by `mask = (mask << 1) + 1`  index.cpp:182-215
  * chuck_size = max(4096, size / 256)                                                      index.cpp:218
  * a bucket of more than chuck_size entries is split by `character()`: 0 for "suffix ends here", else the byte as a
    SIGNED char minus CHAR_MIN plus 1 (so 0x80..0xFF come before 0x00..0x7F); children are handled at offset + 1 and
    the end-of-document child is final                                                      index.cpp:96-126, index.h:66-73
  * smaller buckets are sorted by the rest of the suffix in unsigned byte order             index.cpp:86-95
  * the two bisections of query(), the sort of the hit documents and their run lengths      index.cpp:237-326
Runs of EQUAL suffixes come out of the reference in an order that depends on its in-place swaps and on std::sort; like
the oracle's canonicalize() this model orders them by document index.
"""


def _grow(limit):
    m = 1
    while m < limit:
        m = (m << 1) + 1
    return m


class RefModel:
    def __init__(self, ids, docs):
        self.ids = list(ids)
        self.docs = [bytes(d) for d in docs]
        mask1 = _grow(len(self.docs))
        mask2 = 1
        for d in self.docs:
            while mask2 < len(d):
                mask2 = (mask2 << 1) + 1
        self.size = sum(len(d) for d in self.docs)
        self.bits = bin(mask1).count("1")
        self.mask = mask1
        self.width = 4 if self.bits + bin(mask2).count("1") <= 32 else 8
        self.sa = [(j << self.bits) | i for i, d in enumerate(self.docs) for j in range(len(d))]
        self.chuck = max(4096, self.size // 256)
        self._sort()

    def _rest(self, e, off):
        return self.docs[e & self.mask][(e >> self.bits) + off:]

    def _symbol(self, e, off):
        d = self.docs[e & self.mask]
        p = (e >> self.bits) + off
        if p == len(d):
            return 0
        b = d[p]
        return (b - 256 if b >= 128 else b) + 128 + 1

    def _sort(self):
        work = [(0, self.size, 0)]
        while work:
            lo, hi, off = work.pop()
            if hi - lo <= self.chuck:
                self.sa[lo:hi] = sorted(self.sa[lo:hi], key=lambda e: (self._rest(e, off), e & self.mask))
                continue
            kids = {}
            for e in self.sa[lo:hi]:
                kids.setdefault(self._symbol(e, off), []).append(e)
            at = lo
            for sym in sorted(kids):
                chunk = kids[sym]
                if sym == 0:
                    chunk = sorted(chunk, key=lambda e: e & self.mask)  # equal suffixes: canonical order
                self.sa[at:at + len(chunk)] = chunk
                if sym != 0:
                    work.append((at, at + len(chunk), off + 1))
                at += len(chunk)

    def query(self, kw):
        if not kw:
            raise RuntimeError("Empty keywords are not allowed")
        kw = bytes(kw)
        full = lambda m: self._rest(self.sa[m], 0)
        lo, hi = 0, self.size - 1
        while lo < hi:
            mid = lo + (hi - lo) // 2
            if kw <= full(mid):
                hi = mid
            else:
                lo = mid + 1
        left = lo
        lo, hi = left - 1, self.size - 1
        while lo < hi:
            mid = lo + (hi - lo + 1) // 2
            if full(mid)[:len(kw)] == kw:
                lo = mid
            else:
                hi = mid - 1
        right = lo + 1
        rows = []
        if left < right:
            hit = sorted(self.sa[i] & self.mask for i in range(left, right))
            start = 0
            for i in range(1, len(hit) + 1):
                if i == len(hit) or hit[i] != hit[start]:
                    rows.append((self.ids[hit[start]], i - start))
                    start = i
        return rows
