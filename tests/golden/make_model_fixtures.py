"""Writes tests/golden/model_cases.json: SHA-256 of the canonical suffix array and the rows of sampled keywords for the
fixture classes (3)-(6) of SURVEY.md §8(c), computed by the pure-Python second restatement (tests/ref_model.py).
The reference itself cannot be built in this image (its sources need <format>), so these are NOT outputs of the
reference: they pin the C oracle and the GPU path against an independent reading of index.cpp.
usage: python tests/golden/make_model_fixtures.py        (a few minutes of pure Python)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from coffeedb_amd import workloads as W  # noqa: E402
from ref_model import RefModel  # noqa: E402


def cases():
    """name -> (blob, doc_start, ids, keyword sampling)"""
    out = {}
    # (3) radix path: more than chuck_size = 4096 suffixes; three symbols make several levels of radix nodes
    blob, ds = W.ascii_corpus(100, 400, seed=41, lo=0x61, hi=0x63)
    out["radix_multilevel_abc"] = (blob, ds, np.arange(100, dtype=np.int64) * 7 - 3)
    blob, ds = W.ragged_corpus(600, 40, seed=42, empty_every=7)
    out["radix_ragged_az"] = (blob, ds, np.arange(600, dtype=np.int64) + 1000)
    # (5) bytes >= 0x80: signed child order inside radix nodes, unsigned order inside leaves
    blob, ds = W.ascii_corpus(80, 250, seed=43, lo=0x00, hi=0xFF)
    out["high_bytes_uniform"] = (blob, ds, np.arange(80, dtype=np.int64))
    raw = W.random_bytes(30000, 44, 0, 3)
    blob = np.array([0x10, 0x7F, 0x80, 0xF0], dtype=np.uint8)[raw]
    ds = (np.arange(61) * 500).astype(np.uint64)
    out["high_bytes_four_symbols_deep"] = (blob, ds, np.arange(60, dtype=np.int64) * 2)
    # (6) width boundary: bits1 + bits2 == 32 (u32) and == 33 (u64)
    lens = np.full(1 << 15, 1, dtype=np.uint64); lens[7] = 40000
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    out["width_32_bits_u32"] = (W.random_bytes(int(ds[-1]), 45, 0x30, 0x39), ds, np.arange(1 << 15, dtype=np.int64))
    lens = np.full(1 << 15, 1, dtype=np.uint64); lens[9] = 70000
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    out["width_33_bits_u64"] = (W.random_bytes(int(ds[-1]), 46, 0x30, 0x39), ds, np.arange(1 << 15, dtype=np.int64))
    return out


def keywords(blob, ds, seed):
    pb, po = W.sample_patterns(blob, ds, 60, 1, 6, seed=seed, miss_frac=0.2)
    return [bytes(pb[int(po[i]):int(po[i + 1])]) for i in range(len(po) - 1)]


def model_of(blob, ds, ids):
    docs = [bytes(blob[int(ds[i]):int(ds[i + 1])]) for i in range(len(ds) - 1)]
    return RefModel(ids.tolist(), docs)


def sa_hash(sa, width):
    return hashlib.sha256(np.asarray(sa, dtype=np.uint32 if width == 4 else np.uint64).tobytes()).hexdigest()


if __name__ == "__main__":
    doc = {"_provenance": "tests/golden/make_model_fixtures.py: pure-Python restatement tests/ref_model.py (NOT the reference binary, "
                          "which cannot be built in this image); corpora from coffeedb_amd.workloads with the seeds in the script",
           "cases": {}}
    for k, (name, (blob, ds, ids)) in enumerate(cases().items()):
        m = model_of(blob, ds, ids)
        kws = keywords(blob, ds, 100 + k)
        doc["cases"][name] = {"size": m.size, "bits": m.bits, "mask": m.mask, "width": m.width, "chuck_size": m.chuck,
                              "sa_sha256": sa_hash(m.sa, m.width),
                              "queries": {kw.hex(): m.query(kw) for kw in kws}}
        print(name, m.size, m.bits, m.width, doc["cases"][name]["sa_sha256"][:16], flush=True)
    with open(os.path.join(HERE, "model_cases.json"), "w") as f:
        json.dump(doc, f, indent=1)
