#!/usr/bin/env python3
"""Scans gfx950 assembly (hipcc -S --cuda-device-only) for the miscompile behind round 5's wrong sweep records.

The compiler keeps wave-uniform conditions (template-free switches such as `2 * q < ns1`, `part_m > 1`) as LANE MASKS and, around
inlined code with many of them, re-materialises a mask with vector instructions under the CURRENT exec:
    v_cndmask_b32 vN, 0, 1, s[c:c+1]   ;   v_cmp_ne_u32_e64 s[a:a+1], 1, vN        (bits only for the lanes active HERE)
Inside a loop that lanes leave at different times (s_andn2_b64 exec, exec, sM ; s_cbranch_exec*) the last iteration defines the
mask for the lanes still running; code BEHIND the loop (s_or_b64 exec, exec, sM) that tests the same register pair
    s_and_b64 vcc, exec, s[a:a+1] ; s_cbranch_vccz ...
then runs lanes that had left earlier against zeros — "condition true" for all of them.  rs_sweep_records_kernel's paired loop
("for (; p + NT < kept; p += 2 NT) { rec(p); rec(p + NT); }  if (p < kept) rec(p);") did exactly that for the tail's lanes.

Rule: a definition of that form in a block of a DIVERGENT loop (block comments "in Loop: Header=...", a latch that narrows exec)
whose register pair is tested in a block outside that loop, before it is written again (layout order).  The buggy build is flagged
(4 uses); the product's current objects must scan clean: `make -C coffeedb_amd/csrc isa-scan`.  usage: isa_lanemask_scan.py file.s ..."""
import re, sys, bisect
DEF = re.compile(r'^\s*v_cmp_ne_u32_e64 (s\[\d+:\d+\]), 1, v\d+')
USE = re.compile(r'^\s*s_(and|andn2)_b64 vcc, exec, (s\[\d+:\d+\])')
LATCH = re.compile(r'^\s*s_andn2_b64 exec, exec, s\[\d+:\d+\]')
FUNC = re.compile(r'^(_Z[\w$.]+):')
BLOCK = re.compile(r'^(\.LBB\d+_\d+):(.*)$')
HDR_IN = re.compile(r'in Loop: Header=(BB\d+_\d+)')
HDR_SELF = re.compile(r'=>This (Inner )?Loop Header')


def written(line, rs):
    m = re.match(r'^\s*([sv]_\w+)\s+(.*)$', line)
    if not m or USE.match(line): return False
    ops = [o.strip() for o in m.group(2).split(',')]
    for o in (ops[:2] if m.group(1).startswith('v_') else ops[:1]):
        mm = re.match(r'^s\[(\d+):(\d+)\]$', o) or re.match(r'^s(\d+)$', o)
        if mm:
            g = mm.groups(); lo = int(g[0]); hi = int(g[1]) if len(g) > 1 else lo
            if rs & set(range(lo, hi + 1)): return True
    return False


total = 0
for path in sys.argv[1:]:
    lines = open(path, errors='replace').read().split('\n')
    funcs = [(i, m.group(1)) for i, l in enumerate(lines) if (m := FUNC.match(l))]
    fstarts = [f[0] for f in funcs]
    loop_of = [None] * len(lines)   # innermost loop header of the block every line sits in
    cur = None
    for i, l in enumerate(lines):
        if FUNC.match(l): cur = None
        b = BLOCK.match(l)
        if b:
            c = b.group(2)
            if HDR_SELF.search(c): cur = b.group(1)[2:]          # ".LBB180_239" -> "BB180_239"
            else:
                h = HDR_IN.search(c)
                cur = h.group(1) if h else None
        if '.Lfunc_end' in l: cur = None
        loop_of[i] = cur
    divergent = {loop_of[i] for i, l in enumerate(lines) if LATCH.match(l) and loop_of[i]}
    for i, l in enumerate(lines):
        m = DEF.match(l)
        if not m or loop_of[i] not in divergent: continue
        R = m.group(1); a, b = map(int, re.match(r's\[(\d+):(\d+)\]', R).groups()); rs = set(range(a, b + 1))
        k = bisect.bisect_right(fstarts, i) - 1
        end = fstarts[k + 1] if k + 1 < len(fstarts) else len(lines)
        for j in range(i + 1, end):
            lj = lines[j]
            if '.Lfunc_end' in lj: break
            u = USE.match(lj)
            if u and u.group(2) == R and loop_of[j] != loop_of[i]:
                print(f'{path}:{j + 1}: {funcs[k][1][:80]}: {R} defined at line {i + 1} inside divergent loop {loop_of[i]}, tested outside it here')
                total += 1
            if written(lj, rs): break
print('suspicious uses:', total)
sys.exit(1 if total else 0)
