#!/usr/bin/env python3
"""Text / JSON summaries of rocprofv3 CSV output for profiles/r04 (round 4).
  r04_summary.py stats <dir>      kernel stats (rocprofv3 --stats rows of the mrp_ kernels) and, per kernel and grid, calls / avg / min / max
                                  of the replay leg's launches (kernels whose grid repeats at least 20 times)
  r04_summary.py traffic <fetch_dir> <write_dir> <chunks> <launches> <out.json> <label> <kernel source file>
                                  HBM bytes per replay launch of mrp_sweep_i32_kernel = sum over its size classes of
                                  (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 tallies a wide coalesced read at one half,
                                  MI355X_MICROARCH.md "HBM").  The two counter passes run bench.py with --steps 0 --warmup 0, so the only
                                  sweeps in the trace besides the replay's are those of the recording phase (one chunk at a time: small
                                  grids, other call counts); the replay's classes are the dispatch groups with EXACTLY <launches> calls
                                  (its warm-up launches + --roofline-steps), and the file lists them so that a stray group would show."""
import collections, csv, glob, hashlib, json, sys


def rows_of(d, pattern):
    out = []
    for f in glob.glob(d + "/**/" + pattern, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def short(n):
    return n.split("(")[0].replace("void ", "")


def stats(d):
    st = rows_of(d, "*kernel_stats.csv")
    if st:
        print("rocprofv3 --stats (kernels of libmargin_rphmm.so):")
        for r in st:
            if "mrp_" in r["Name"] or "phm_" in r["Name"]:
                print(f"  {short(r['Name']):44s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs']) / 1e6:10.3f} ms  avg {float(r['AverageNs']) / 1e6:8.4f} ms  {float(r['Percentage']):5.1f} %")
    by = collections.defaultdict(list)
    for r in rows_of(d, "*kernel_trace.csv"):
        if "mrp_" in r["Kernel_Name"]:
            by[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("dispatch groups with at least 20 calls of the same grid (the replay leg's launches):")
    for (name, g, wg), v in sorted(by.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
        if g >= 100000 and len(v) >= 20:
            print(f"  {name:44s} grid {g:>10d} wg {wg:>4s} calls {len(v):3d} avg {sum(v) / len(v):8.3f} ms  min {min(v):8.3f}  max {max(v):8.3f}")


def pmc_table(d):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows_of(d, "*counter_collection.csv"):
        if "mrp_" in r["Kernel_Name"]:
            by[(short(r["Kernel_Name"]), int(r["Grid_Size"]), r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return by


def traffic(fd, wd, chunks, launches, out, where, kernel_file):
    f, w = pmc_table(fd), pmc_table(wd)
    launches = int(launches)
    tot, detail, others = 0.0, [], []
    for key, dd in sorted(f.items(), key=lambda kv: -kv[0][1]):
        name, g, wg = key
        if name != "mrp_sweep_i32_kernel":
            continue
        n = len(dd["FETCH_SIZE"])
        if n != launches or g < 100000:
            others.append(dict(grid=g, workgroup=int(wg), calls=n))
            continue
        fe = sum(dd["FETCH_SIZE"]) / n
        wr = sum(w[key]["WRITE_SIZE"]) / len(w[key]["WRITE_SIZE"])
        b = (2.0 * fe + wr) * 1024.0
        detail.append(dict(grid=g, workgroup=int(wg), launches=n, FETCH_SIZE_KB=fe, WRITE_SIZE_KB=wr, hbm_bytes=b))
        tot += b
    sha = hashlib.sha256(open(kernel_file, "rb").read()).hexdigest()
    json.dump(dict(chunks=int(chunks), sweep_kernel_hbm_bytes_per_launch=tot, kernel_source_sha256=sha,
                   source=where + ": rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps 0 --warmup 0` (replay leg only); "
                          "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the size classes of one replay launch",
                   replay_launches_per_class=launches, classes=detail,
                   other_sweep_dispatch_groups_not_counted=sorted(others, key=lambda o: -o["grid"])[:12]), open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    {"stats": lambda: stats(sys.argv[2]), "traffic": lambda: traffic(*sys.argv[2:9])}[sys.argv[1]]()
