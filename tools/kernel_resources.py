"""Summarises hipcc -Rpass-analysis=kernel-resource-usage output: registers, spills, LDS and occupancy per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt ; python tools/kernel_resources.py res.txt [filter]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in t.split("Function Name: ")[1:]:
    mangled = b.split()[0]
    if flt and flt not in mangled:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    m = re.search(r"RsCfgI([^E]*(?:E[^E]*?)*?)EEE?NS", mangled)
    cfg = re.findall(r"L[ib](\d+)E", mangled.split("RsCfgI")[1].split("EEN")[0]) if "RsCfgI" in mangled else []
    head = mangled.split("RsCfg")[0][-12:] if cfg else mangled[:60]
    gen = "TextGen" in mangled
    tail = mangled.split("TextGenE" if gen else "NoGenE")[-1][:2] if cfg else ""
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{head:>14} cfg={','.join(cfg):<34} {'gen' if gen else '   '} w={tail:<3} VGPR {g('VGPRs'):3d} AGPR {g('AGPRs'):3d} scratch {scratch:4d} "
          f"spill {g('VGPRs Spill'):3d} occ {occ} LDS {lds}")
