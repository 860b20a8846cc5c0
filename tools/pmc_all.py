#!/usr/bin/env python3
"""Per kernel and grid: mean of each rocprofv3 counter per dispatch, every mrp_ kernel (no replay-only filter), with the derived
ratios the round-4 questions need.  Usage: pmc_all.py <rocprofv3 output dir> [min grid]"""
import collections, csv, glob, sys


def short(n):
    return n.split("(")[0].replace("void ", "")


def main():
    d = sys.argv[1]
    min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mrp_" in r["Kernel_Name"]:
                by[(short(r["Kernel_Name"]), int(r["Grid_Size"]), r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("counters, mean per dispatch (dispatches of the same kernel, grid and workgroup size pooled):")
    for (name, g, wg), dd in sorted(by.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
        if g < min_grid:
            continue
        n = len(next(iter(dd.values())))
        m = {c: sum(v) / len(v) for c, v in dd.items()}
        line = f"  {name:34s} grid {g:>10d} wg {wg:>4s} n={n:3d}: " + ", ".join(f"{c}={v:.5g}" for c, v in sorted(m.items()))
        extra = []
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
                if c in m:
                    extra.append(f"{c}/WAVE_CYCLES={m[c] / wc:.3f}")
        if m.get("SQ_INSTS_LDS") and "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
            extra.append(f"LDS_BANK_CONFLICT/LDS_IDX_ACTIVE={m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
        if "FETCH_SIZE" in m:
            extra.append(f"fetch_bytes(2x)={2 * 1024 * m['FETCH_SIZE']:.4g}")
        if "WRITE_SIZE" in m:
            extra.append(f"write_bytes={1024 * m['WRITE_SIZE']:.4g}")
        print(line + ("  | " + ", ".join(extra) if extra else ""))


if __name__ == "__main__":
    main()
