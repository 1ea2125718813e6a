#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits.  For every scratch_load / scratch_store of a gfx950 kernel (hipcc -S output) the SMALLEST loop
(label ... backward branch) that contains it, with that loop's length and MFMA count: a spill in a loop with MFMAs and a short body is
paid per contraction step, one in a long outer loop or outside every loop once per slab / per workgroup.
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -S --cuda-device-only csrc/<file>.hip -o <file>.s
    tools/scratch_loops.py <file>.s '<mangled-name regex>' ..."""
import collections
import re
import subprocess
import sys


def kernels(lines):
    start = None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            start = (m.group(1), i)
        elif start and l.strip().startswith('s_endpgm'):
            yield start[0], start[1], i
            start = None


def main():
    lines = open(sys.argv[1]).read().split('\n')
    pats = [re.compile(p) for p in sys.argv[2:]]
    for name, a, b in kernels(lines):
        if pats and not any(p.search(name) for p in pats):
            continue
        body = lines[a:b + 1]
        nice = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '')
        scr = [i for i, x in enumerate(body) if 'scratch_' in x]
        mf = [i for i, x in enumerate(body) if 'v_mfma' in x]
        print(f"{nice[:160]}\n  {b - a} lines, {len(mf)} MFMA instructions, {len(scr)} scratch instructions")
        if not scr:
            continue
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
        loops = set()
        for i, l in enumerate(body):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and labels.get(m.group(1), i) < i:
                loops.add((labels[m.group(1)], i))
        where = collections.Counter()
        for i in scr:
            inside = [(e - s, s, e) for s, e in loops if s <= i <= e]
            where[min(inside)[1:] if inside else None] += 1
        for key, n in sorted(where.items(), key=lambda kv: (kv[0] is None, kv[0])):
            if key is None:
                print(f"    {n:3d} outside every loop (once per workgroup)")
            else:
                s, e = key
                m_in = sum(s <= i <= e for i in mf)
                print(f"    {n:3d} in the loop at lines {s}-{e}: {e - s} instructions long, {m_in} MFMAs")


if __name__ == '__main__':
    main()
