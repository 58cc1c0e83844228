"""Quick on-GPU probe: build + query timings at a few sizes, with per-kernel HIP-event times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coffeedb_amd import capi, workloads as W

def run(nd, dl, npat, check=False, **opts):
    t = time.time(); blob, ds = W.ascii_corpus(nd, dl, seed=12345); tg = time.time() - t
    ids = np.arange(nd, dtype=np.int64)
    g = capi.GpuStringIndex(); g.set_option("profile", 1)
    for k, v in opts.items(): g.set_option(k, v)
    g.add_bulk(ids, blob, ds)
    for rep in range(2):
        g.profile_reset(); t = time.time(); g.build(); tb = time.time() - t
    n = g.size
    print(f"n={n} ({n/2**20:.0f} MiB) gen {tg:.1f}s build wall {tb*1e3:.1f} ms lib {g.stat('build_ms'):.1f} ms -> {n/2**30/tb:.2f} GiB/s "
          f"rounds={g.stat('rounds'):.0f} ext={g.stat('ext_rounds'):.0f} dbl={g.stat('dbl_rounds'):.0f} unres0={g.stat('unresolved_after_initial'):.0f} "
          f"passes={g.stat('sort_passes'):.0f} skipped={g.stat('sort_passes_skipped'):.0f} nsym={g.stat('key_symbols'):.0f} symbits={g.stat('symbol_bits'):.0f}", flush=True)
    for k, v in sorted(g.profile().items(), key=lambda kv: -kv[1]["ms"]):
        gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
        print(f"   {k:24s} {v['ms']:9.3f} ms  x{v['launches']:<4d} {gbs:8.1f} GB/s")
    pb, po = W.sample_patterns(blob, ds, npat, 4, 16, seed=5)
    for rep in range(3):
        g.profile_reset(); t = time.time(); rp, gi, gc, hits = g.query_batch(pb, po); tq = time.time() - t
    print(f"   query {npat} patterns: {tq*1e3:.2f} ms -> {npat/tq/1e6:.2f} M patterns/s, hits={hits} rows={len(gi)}")
    for k, v in sorted(g.profile().items(), key=lambda kv: -kv[1]["ms"]):
        print(f"   {k:24s} {v['ms']:9.3f} ms  x{v['launches']}")
    if check:
        from oracle import OracleIndex
        o = OracleIndex(); o.add_bulk(ids, blob, ds); t = time.time(); o.build(); to = time.time() - t; o.canonicalize()
        print(f"   oracle build {to:.2f}s; SA equal: {np.array_equal(g.sa(), o.sa())}")
        t = time.time(); orp, oi, oc, oh = o.query_batch(pb, po, nthreads=1); tq1 = time.time() - t
        print(f"   oracle query 1 thread {npat/tq1/1e3:.0f} k/s; rows equal: {np.array_equal(gi, oi) and np.array_equal(gc, oc) and np.array_equal(rp, orp)}")
    g.close()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    run(10000, 256, 1000, check=True)
    run(1 << 16, 1024, 100000, check=(which == "check"))
    if which in ("big", "check"):
        run(1 << 18, 1024, 100000)
        run(1 << 20, 1024, 100000)
