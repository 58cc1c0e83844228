"""Deterministic synthetic corpora and pattern batches.

The reference's own tests draw from an unseeded `random` (test/test-string.py:26), so they are not
reproducible; these generators use a counter-based splitmix64 so that every test, the bench and the
CPU baseline see byte-identical inputs for a given (shape, seed).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser over a uint64 array of counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _draws(count, seed, stream):
    base = np.uint64((seed * 0x100000001B3 + stream * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF)
    out = np.empty(count, dtype=np.uint64)
    step = 1 << 22
    for s in range(0, count, step):
        e = min(count, s + step)
        with np.errstate(over="ignore"):
            out[s:e] = splitmix64(np.arange(s, e, dtype=np.uint64) + base)
    return out


def random_bytes(n, seed, lo=0x20, hi=0x7E, stream=0):
    """n bytes uniform over [lo, hi] (8 bytes per 64-bit draw, scaled from 0..255)."""
    span = hi - lo + 1
    out = np.empty(((n + 7) // 8) * 8, dtype=np.uint8)
    step = 1 << 22  # draws per chunk
    ndraw = len(out) // 8
    base = np.uint64((seed * 0x100000001B3 + stream * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF)
    for s in range(0, ndraw, step):
        e = min(ndraw, s + step)
        with np.errstate(over="ignore"):
            r = splitmix64(np.arange(s, e, dtype=np.uint64) + base)
        b = r.view(np.uint8).reshape(-1, 8).astype(np.uint16)
        out[s * 8:e * 8] = (lo + ((b * span) >> 8)).astype(np.uint8).reshape(-1)
    return out[:n]


def uniform_docs(ndocs, doclen):
    return np.arange(ndocs + 1, dtype=np.uint64) * np.uint64(doclen)


def ascii_corpus(ndocs, doclen, seed=12345, lo=0x20, hi=0x7E):
    """C0/C1-style corpus: ndocs documents of exactly doclen bytes, uniform over [lo, hi]."""
    return random_bytes(ndocs * doclen, seed, lo, hi), uniform_docs(ndocs, doclen)


def ragged_corpus(ndocs, maxlen, seed=7, lo=0x61, hi=0x7A, empty_every=0):
    """Documents of varying length in [0, maxlen] (optionally every k-th one empty)."""
    lens = (_draws(ndocs, seed, 1) % np.uint64(maxlen + 1)).astype(np.uint64)
    if empty_every:
        lens[::empty_every] = 0
    doc_start = np.zeros(ndocs + 1, dtype=np.uint64)
    np.cumsum(lens, out=doc_start[1:])
    return random_bytes(int(doc_start[-1]), seed, lo, hi), doc_start


def zipf_corpus(ndocs, doclen, seed=2, nsym=64, base=0x30):
    """C2-style skewed alphabet: nsym symbols base+rank with P(rank k) ∝ 1/k (k = 1..nsym)."""
    w = 1.0 / np.arange(1, nsym + 1)
    cdf = np.cumsum(w / w.sum())
    n = ndocs * doclen
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 22
    for s in range(0, n, step):
        e = min(n, s + step)
        u = (_draws(e - s, seed, 3 + s // step) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        out[s:e] = (base + np.searchsorted(cdf, u, side="right").clip(0, nsym - 1)).astype(np.uint8)
    return out, uniform_docs(ndocs, doclen)


def utf8_corpus(ndocs, approx_doclen, seed=4):
    """C4-style valid UTF-8: 50 % 1-byte, 30 % 2-byte (U+0080–07FF), 20 % 3-byte (U+0800–FFFF minus
    surrogates) code points; documents are cut at code-point boundaries near approx_doclen bytes."""
    parts, doc_start = [], [0]
    total = 0
    for d in range(ndocs):
        ncp = max(1, int(approx_doclen / 1.7))
        r = _draws(ncp * 2, seed, 10 + d)
        cls = (r[:ncp] % np.uint64(10)).astype(np.int64)
        val = r[ncp:]
        cp = np.where(cls < 5, 0x20 + (val % np.uint64(0x5F)).astype(np.int64),
                      np.where(cls < 8, 0x80 + (val % np.uint64(0x780)).astype(np.int64),
                               0x800 + (val % np.uint64(0xD000)).astype(np.int64)))  # < 0xD800
        s = "".join(map(chr, cp.tolist())).encode("utf-8")
        parts.append(np.frombuffer(s, dtype=np.uint8))
        total += len(s)
        doc_start.append(total)
    blob = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    return blob, np.asarray(doc_start, dtype=np.uint64)


def sample_patterns(blob, doc_start, npat, mmin=4, mmax=16, seed=99, miss_frac=0.1, miss_byte=None):
    """Pattern batch: substrings sampled at uniform (doc, offset) — guaranteed ≥1 hit — plus a
    miss_frac tail of patterns whose last byte is replaced so that most of them do not occur.
    Returns (pattern_blob uint8[], offsets uint64[npat+1])."""
    nd = len(doc_start) - 1
    r = _draws(npat * 4, seed, 2)
    lens = doc_start[1:] - doc_start[:-1]
    nonempty = np.nonzero(lens > 0)[0]
    assert len(nonempty) > 0
    docs = nonempty[(r[:npat] % np.uint64(len(nonempty))).astype(np.int64)]
    m = (mmin + (r[npat:2 * npat] % np.uint64(mmax - mmin + 1))).astype(np.int64)
    dl = lens[docs].astype(np.int64)
    m = np.minimum(m, dl)
    off = (r[2 * npat:3 * npat] % (dl - m + 1).astype(np.uint64)).astype(np.int64)
    start = doc_start[docs].astype(np.int64) + off
    offsets = np.zeros(npat + 1, dtype=np.uint64)
    np.cumsum(m.astype(np.uint64), out=offsets[1:])
    idx = np.repeat(start - offsets[:-1].astype(np.int64), m) + np.arange(int(offsets[-1]), dtype=np.int64)
    pblob = blob[idx].copy()
    nmiss = int(npat * miss_frac)
    if nmiss:
        which = (r[3 * npat:3 * npat + nmiss] % np.uint64(npat)).astype(np.int64)
        last = offsets[which + 1].astype(np.int64) - 1
        if miss_byte is None:
            pblob[last] = np.where(pblob[last] == 0x7E, 0x21, pblob[last] + 1).astype(np.uint8)
        else:
            pblob[last] = miss_byte
    return pblob, offsets


# ---- the same byte stream generated with torch ops (on the GPU for the bench; bit-identical to
# random_bytes so a host-side prefix can be regenerated with numpy for the CPU baseline) ----------
def _torch_lsr(x, k):
    import torch
    return (x >> k) & torch.tensor((1 << (64 - k)) - 1, dtype=torch.int64, device=x.device)


def _as_i64(v):
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def random_bytes_torch(n, seed, lo=0x20, hi=0x7E, stream=0, device="cpu", chunk_draws=1 << 24):
    import torch
    span = hi - lo + 1
    ndraw = (n + 7) // 8
    out = torch.empty(ndraw * 8, dtype=torch.uint8, device=device)
    base = _as_i64(seed * 0x100000001B3 + stream * 0x9E3779B1)
    c0, c1, c2 = _as_i64(0x9E3779B97F4A7C15), _as_i64(0xBF58476D1CE4E5B9), _as_i64(0x94D049BB133111EB)
    for s in range(0, ndraw, chunk_draws):
        e = min(ndraw, s + chunk_draws)
        z = torch.arange(s, e, dtype=torch.int64, device=device) + base + c0
        z = (z ^ _torch_lsr(z, 30)) * c1
        z = (z ^ _torch_lsr(z, 27)) * c2
        z = z ^ _torch_lsr(z, 31)
        b = z.view(torch.uint8).to(torch.int32)
        out[s * 8:e * 8] = (lo + ((b * span) >> 8)).to(torch.uint8)
    return out[:n]


def zipf_bytes_torch(n, seed=2, nsym=64, base=0x30, device="cuda", chunk=1 << 28):
    """C2 text generated on the device: nsym symbols base+rank with P(rank k) ∝ 1/k (torch's own generator:
    deterministic for a given seed and torch build, not bit-identical to zipf_corpus)."""
    import torch
    w = 1.0 / torch.arange(1, nsym + 1, dtype=torch.float64, device=device)
    cdf = torch.cumsum(w / w.sum(), 0)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    gen = torch.Generator(device=device).manual_seed(seed)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        u = torch.rand(e - s, dtype=torch.float64, device=device, generator=gen)
        out[s:e] = (base + torch.searchsorted(cdf, u).clamp_(0, nsym - 1)).to(torch.uint8)
    return out


def utf8_bytes_torch(total_bytes, seed=4, device="cuda", doc_bytes=1024, step=1 << 27):
    """C4-style valid UTF-8 generated on the device: 50 % 1-byte, 30 % 2-byte (U+0080-07FF), 20 % 3-byte
    (U+0800-D7FF) code points; documents are cut at the first code-point boundary at or behind every multiple of
    doc_bytes.  Returns (text uint8[n] with n a multiple of 16, doc_start uint64 numpy[ndocs + 1])."""
    import torch
    ncp = int(total_bytes / 1.7)
    g_ = torch.Generator(device=device).manual_seed(seed)
    cuts = []
    total = 0
    buf = torch.empty(int(ncp * 1.72) + (1 << 20), dtype=torch.uint8, device=device)
    for s in range(0, ncp, step):
        m = min(step, ncp - s)
        cls = torch.randint(0, 10, (m,), device=device, generator=g_)
        val = torch.randint(0, 1 << 30, (m,), device=device, generator=g_)
        cp = torch.where(cls < 5, 0x20 + val % 0x5F, torch.where(cls < 8, 0x80 + val % 0x780, 0x800 + val % 0xD000))
        ln = torch.where(cp < 0x80, 1, torch.where(cp < 0x800, 2, 3))
        off = torch.cumsum(ln, 0) - ln
        nb = int((off[-1] + ln[-1]).item())
        out = buf[total:total + nb]
        one, two, three = cp < 0x80, (cp >= 0x80) & (cp < 0x800), cp >= 0x800
        out[off[one]] = cp[one].to(torch.uint8)
        out[off[two]] = (0xC0 | (cp[two] >> 6)).to(torch.uint8)
        out[off[two] + 1] = (0x80 | (cp[two] & 0x3F)).to(torch.uint8)
        out[off[three]] = (0xE0 | (cp[three] >> 12)).to(torch.uint8)
        out[off[three] + 1] = (0x80 | ((cp[three] >> 6) & 0x3F)).to(torch.uint8)
        out[off[three] + 2] = (0x80 | (cp[three] & 0x3F)).to(torch.uint8)
        first = (total + doc_bytes - 1) // doc_bytes * doc_bytes
        tg = torch.arange(first, total + nb, doc_bytes, device=device) - total
        idx = torch.searchsorted(off, tg).clamp_(max=m - 1)
        cuts.append((off[idx] + total).cpu().numpy())
        total += nb
        del cls, val, cp, ln, off, one, two, three, tg, idx
    # the corpus ends at the last document cut that keeps the length a multiple of 16 (device text must be 16-byte
    # aligned for the next shard behind it; a cut is a code-point boundary, so the text stays valid UTF-8)
    cuts = np.concatenate(cuts) if cuts else np.zeros(0, dtype=np.int64)
    cuts = cuts[cuts <= total]
    ok = cuts[cuts % 16 == 0]
    end = int(ok[-1]) if len(ok) else 0
    ds = np.unique(np.concatenate([[0], cuts[cuts < end], [end]])).astype(np.uint64)
    return buf[:end], ds


def sample_patterns_torch(text, doc_start, npat, mmin, mmax, seed=99, miss_frac=0.1, miss_byte=0x7F, utf8=False):
    """Device-side pattern batch for corpora that have no host copy: substrings at uniform (document, offset),
    lengths uniform in [mmin, mmax] (clipped to the document), a miss_frac share with the last byte replaced by a
    byte the corpus never holds.  utf8=True moves both ends forward to code-point boundaries.
    doc_start: uint64 numpy / int64 tensor [ndocs + 1].  Returns (blob uint8 tensor, offsets int64 tensor, nbytes)."""
    import torch
    dev = text.device
    ds = torch.as_tensor(np.asarray(doc_start).astype(np.int64)) if not torch.is_tensor(doc_start) else doc_start
    ds = ds.to(dev)
    nd = ds.numel() - 1
    g = torch.Generator(device=dev).manual_seed(seed)
    docs = torch.randint(0, nd, (npat,), device=dev, generator=g)
    lens = ds[docs + 1] - ds[docs]
    m = torch.randint(mmin, mmax + 1, (npat,), device=dev, generator=g)
    m = torch.minimum(m, lens).clamp_(min=1)
    r = torch.randint(0, 1 << 40, (npat,), device=dev, generator=g)
    start = ds[docs] + r % (lens - m + 1).clamp_(min=1)
    end = start + m
    if utf8:
        dend = ds[docs + 1]
        for _ in range(3):  # skip continuation bytes (10xxxxxx): at most two in a row for <= 3-byte code points
            cont = (start < dend) & ((text[start.clamp(max=text.numel() - 1)] & 0xC0) == 0x80)
            start = start + cont.to(start.dtype)
        end = torch.maximum(start + 1, torch.minimum(start + m, dend))
        for _ in range(3):
            cont = (end < dend) & ((text[end.clamp(max=text.numel() - 1)] & 0xC0) == 0x80)
            end = end + cont.to(end.dtype)
        keep = start < dend
        start = torch.where(keep, start, ds[docs])  # (a document whose tail is all continuation bytes cannot occur)
        end = torch.where(keep, end, torch.minimum(ds[docs] + 1, dend))
    m = (end - start).clamp_(min=1)
    offs = torch.zeros(npat + 1, dtype=torch.int64, device=dev)
    torch.cumsum(m, 0, out=offs[1:])
    nbytes = int(offs[-1].item())
    idx = torch.repeat_interleave(start - offs[:-1], m) + torch.arange(nbytes, device=dev)
    blob = torch.zeros(nbytes + 16, dtype=torch.uint8, device=dev)
    blob[:nbytes] = text[idx]
    nmiss = int(npat * miss_frac)
    if nmiss:
        which = torch.randint(0, npat, (nmiss,), device=dev, generator=g)
        blob[offs[which + 1] - 1] = miss_byte
    return blob, offs, nbytes
