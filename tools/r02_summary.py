#!/usr/bin/env python3
"""Text summaries of rocprofv3 CSV output for profiles/r02 (round 2).
  r02_summary.py stats <dir>             kernel stats (rocprofv3 --stats rows of the mrp_ kernels) and, per kernel and grid,
                                         calls / avg / min / max of the large dispatches (the replay launches of bench.py)
  r02_summary.py pmc <dir>               per kernel and grid (large dispatches): mean of each counter per dispatch
  r02_summary.py traffic <fetch_dir> <write_dir> <chunks> <out.json> [<label> <kernel source file>]
                                         HBM bytes per replay launch of mrp_sweep_i32_kernel: sum over its size classes of
                                         2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 tallies a wide coalesced read at one half,
                                         MI355X_MICROARCH.md "HBM")"""
import collections, csv, glob, json, sys


def rows_of(d, pattern):
    out = []
    for f in glob.glob(d + "/**/" + pattern, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def short(n):
    return n.split("(")[0].replace("void ", "")


# the kernels of bench.py's replay leg (their launches repeat with identical grids); the kernels of the end-to-end steps are
# listed level by level in pipeline_levels_*.txt instead
REPLAY = {"mrp_sweep_i32_kernel", "mrp_emission_kernel", "mrp_pack_kernel", "mrp_planes_kernel", "mrp_emission_general_kernel"}


def stats(d):
    st = rows_of(d, "*kernel_stats.csv")
    if st:
        print("rocprofv3 --stats (kernels of libmargin_rphmm.so):")
        for r in st:
            if "mrp_" in r["Name"] or "phm_" in r["Name"]:
                print(f"  {short(r['Name']):34s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs']) / 1e6:10.3f} ms  avg {float(r['AverageNs']) / 1e6:8.4f} ms  {float(r['Percentage']):5.1f} %")
    tr = rows_of(d, "*kernel_trace.csv")
    by = collections.defaultdict(list)
    for r in tr:
        if "mrp_" in r["Kernel_Name"]:
            by[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("replay launches (kernels of the replay leg, same grid in every launch: at least 20 calls, grid of at least 100000 work-items):")
    for (name, g, wg), v in sorted(by.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
        if g >= 100000 and 20 <= len(v) <= 64 and name in REPLAY:
            print(f"  {name:34s} grid {g:>10d} wg {wg:>4s} calls {len(v):3d} avg {sum(v) / len(v):8.3f} ms  min {min(v):8.3f}  max {max(v):8.3f}")


def pmc_table(d):
    pm = rows_of(d, "*counter_collection.csv")
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in pm:
        if "mrp_" in r["Kernel_Name"]:
            by[(short(r["Kernel_Name"]), int(r["Grid_Size"]), r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return by


def pmc(d):
    by = pmc_table(d)
    print("counters, mean per dispatch (dispatches with the same kernel and grid, grid of at least 100000 work-items):")
    for (name, g, wg), dd in sorted(by.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
        n = len(next(iter(dd.values())))
        if g >= 100000 and n <= 64 and ((name in REPLAY and n >= 4) or (name == "mrp_cross_emit_kernel" and g > 5000000)):
            print(f"  {name:34s} grid {g:>10d} wg {wg:>4s} n={n:2d}: " + ", ".join(f"{c}={sum(v) / len(v):.5g}" for c, v in sorted(dd.items())))


def traffic(fd, wd, chunks, out, where="profiles/r02", kernel_file=None):
    f, w = pmc_table(fd), pmc_table(wd)
    tot, detail = 0.0, []
    # the replay launches: the call count shared by the kernel's largest grids (warm-up + timed launches); the end-to-end step's
    # sweeps (one call per concurrent batch) have fewer
    counts = sorted(((g, len(dd["FETCH_SIZE"])) for (name, g, wg), dd in f.items() if name == "mrp_sweep_i32_kernel"), reverse=True)
    n_replay = counts[0][1] if counts else 0
    for key, dd in f.items():
        name, g, wg = key
        n = len(dd["FETCH_SIZE"])
        if name != "mrp_sweep_i32_kernel" or g < 100000 or n != n_replay:
            continue
        fe = sum(dd["FETCH_SIZE"]) / n
        wr = sum(w[key]["WRITE_SIZE"]) / len(w[key]["WRITE_SIZE"])
        b = (2.0 * fe + wr) * 1024.0
        detail.append(dict(grid=g, workgroup=int(wg), launches=n, FETCH_SIZE_KB=fe, WRITE_SIZE_KB=wr, hbm_bytes=b))
        tot += b
    import hashlib
    sha = hashlib.sha256(open(kernel_file, "rb").read()).hexdigest() if kernel_file else None
    json.dump(dict(chunks=int(chunks), sweep_kernel_hbm_bytes_per_launch=tot, kernel_source_sha256=sha,
                   source=where + ": rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of bench.py's replay leg; bytes = "
                          "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the kernel's three size classes of one launch",
                   classes=detail), open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    {"stats": lambda: stats(sys.argv[2]), "pmc": lambda: pmc(sys.argv[2]),
     "traffic": lambda: traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], *(sys.argv[6:8]))}[sys.argv[1]]()
