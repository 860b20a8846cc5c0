#!/usr/bin/env python3
"""Names the frames of a cpusampler dump (tools/sampler/cpusampler.c): per sample the leaf symbol, and for leaves outside the
library (libc's memcpy / malloc, the HIP runtime) the first frame inside libmargin_rphmm.so that led there.
usage: resolve.py dump.txt [top N]"""
import bisect, collections, os, subprocess, sys

def symbols(path):
    out = []
    for flags in (["-n", "--defined-only"], ["-D", "-n", "--defined-only"]):
        try:
            txt = subprocess.run(["nm", "-C"] + flags + [path], capture_output=True, text=True).stdout
        except OSError:
            txt = ""
        for line in txt.splitlines():
            p = line.split(None, 2)
            if len(p) == 3 and p[1] in "tTwWiV":
                try: out.append((int(p[0], 16), p[2]))
                except ValueError: pass
    out.sort()
    return [a for a, _ in out], [s for _, s in out]

def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    lines = open(path).read().splitlines()
    n = int(lines[0].split()[1])
    samples = [[int(x, 16) for x in l.split()[1:]] for l in lines[1:1 + n]]
    tids = collections.Counter(l.split()[0] for l in lines[1:1 + n] if l.split())
    maps, base = [], {}
    for l in lines[2 + n:]:
        p = l.split()
        if len(p) < 6: continue
        lo, hi = (int(x, 16) for x in p[0].split("-"))
        off = int(p[2], 16)
        base[p[5]] = min(base.get(p[5], lo - off), lo - off)  # load address (nm prints virtual addresses: the first segment's is 0)
        if "x" in p[1]: maps.append((lo, hi, off, p[5]))
    maps.sort()
    los = [m[0] for m in maps]
    symtab = {}
    def name(addr):
        i = bisect.bisect_right(los, addr) - 1
        if i < 0 or addr >= maps[i][1]: return "?", "?"
        lo, hi, off, mod = maps[i]
        if mod not in symtab: symtab[mod] = symbols(mod) if os.path.exists(mod) else ([], [])
        a, s = symtab[mod]
        rel = addr - base[mod]
        j = bisect.bisect_right(a, rel) - 1
        return os.path.basename(mod), (s[j] if j >= 0 else "?")
    leaf = collections.Counter(); mods = collections.Counter(); ours = collections.Counter(); incl = collections.Counter()
    for fr in samples:
        if not fr: continue
        named = [name(a - (1 if k else 0)) for k, a in enumerate(fr)]
        m, s = named[0]
        leaf[(m, s)] += 1; mods[m] += 1
        seen = set()
        for mm, ss in named:
            if mm.startswith("libmargin") and ss not in seen: incl[ss] += 1; seen.add(ss)
        for mm, ss in named:
            if mm.startswith("libmargin"): ours[ss + ("" if (mm, ss) == named[0] else "   <- " + s[:40])] += 1; break
        else:
            ours["(no library frame)  " + m + ":" + s[:40]] += 1
    tot = sum(leaf.values())
    print(f"{tot} samples on {len(tids)} threads")
    print("\nby module (leaf):")
    for m, c in mods.most_common(12): print(f"  {100.0 * c / tot:5.1f} %  {m}")
    print("\nby leaf symbol:")
    for (m, s), c in leaf.most_common(top): print(f"  {100.0 * c / tot:5.1f} %  {m}: {s[:100]}")
    print("\nby first library frame (<- leaf outside it):")
    for s, c in ours.most_common(top): print(f"  {100.0 * c / tot:5.1f} %  {s[:120]}")
    print("\ninclusive, library functions:")
    for s, c in incl.most_common(top): print(f"  {100.0 * c / tot:5.1f} %  {s[:100]}")

if __name__ == "__main__":
    main()
