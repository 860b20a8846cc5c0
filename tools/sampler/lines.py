#!/usr/bin/env python3
"""Leaf samples of a cpusampler dump inside ONE library, by source line (the library must carry debug info: build it with -g).
Also: leaf samples in libc (memcpy / memset / malloc ...) booked to the calling line inside the library.
usage: lines.py dump.txt /path/to/lib.so [top N]"""
import bisect, collections, os, subprocess, sys

path, lib = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lines = open(path).read().splitlines()
n = int(lines[0].split()[1])
samples = [[int(x, 16) for x in l.split()[1:]] for l in lines[1:1 + n]]
maps, base = [], {}
for l in lines[2 + n:]:
    p = l.split()
    if len(p) < 6: continue
    lo, hi = (int(x, 16) for x in p[0].split("-"))
    off = int(p[2], 16)
    base[p[5]] = min(base.get(p[5], lo - off), lo - off)
    if "x" in p[1]: maps.append((lo, hi, p[5]))
maps.sort()
los = [m[0] for m in maps]
def module(a):
    i = bisect.bisect_right(los, a) - 1
    return maps[i][2] if i >= 0 and a < maps[i][1] else "?"
libname = lib.split("/")[-1]
self_addr, via_addr = collections.Counter(), collections.Counter()
total = 0
for fr in samples:
    if not fr: continue
    total += 1
    body, bm = fr, [module(a) for a in fr]  # (the dump starts at the interrupted pc: handler and trampoline are dropped by the sampler)
    if not body: continue
    if bm[0].endswith(libname): self_addr[body[0] - base[bm[0]]] += 1
    else:
        for a, m in zip(body[1:], bm[1:]):
            if m.endswith(libname): via_addr[(a - 1 - base[m], bm[0].split("/")[-1])] += 1; break
def a2l(addrs):
    """address -> (function, file:line); llvm-symbolizer where there is one (binutils 2.38 does not read the DWARF 5 forms 0x22 / 0x23)"""
    if not addrs: return {}
    sym = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
    if os.path.exists(sym):
        out = subprocess.run([sym, "--obj=" + lib, "-f", "-C", "--no-inlines", "-s"] + [hex(a) for a in addrs], capture_output=True, text=True).stdout.split("\n\n")
        res = {}
        for a, blk in zip(addrs, out):
            l = blk.strip().splitlines()
            res[a] = (l[0], ":".join(l[1].split(":")[:2])) if len(l) >= 2 else ("?", "?")
        return res
    out = subprocess.run(["addr2line", "-e", lib, "-f", "-C", "-s"] + [hex(a) for a in addrs], capture_output=True, text=True).stdout.splitlines()
    return {a: (out[2 * i], out[2 * i + 1]) for i, a in enumerate(addrs)}
res = a2l(list(self_addr))
by_line, by_fn = collections.Counter(), collections.Counter()
for a, c in self_addr.items():
    fn, ln = res.get(a, ("?", "?"))
    by_line[(fn[:50], ln)] += c; by_fn[fn[:60]] += c
print(f"{total} samples; {sum(self_addr.values())} with the leaf inside {libname}, {sum(via_addr.values())} in another module called from it")
print("\nself time by function:")
for k, c in by_fn.most_common(25): print(f"  {100.0 * c / total:5.1f} %  {k}")
print("\nself time by line:")
for (fn, ln), c in by_line.most_common(top): print(f"  {100.0 * c / total:5.1f} %  {ln:28s} {fn}")
res2 = a2l([a for a, _ in via_addr])
by_call = collections.Counter()
for (a, m), c in via_addr.items():
    fn, ln = res2.get(a, ("?", "?"))
    by_call[(fn[:50], ln, m)] += c
print("\ntime in other modules by calling line:")
for (fn, ln, m), c in by_call.most_common(top): print(f"  {100.0 * c / total:5.1f} %  {ln:28s} {fn}  -> {m}")
