#!/usr/bin/env python3
"""Summaries of rocprofv3 CSV output for profiles/r05 (round 5).
  r05_summary.py path_traffic <fetch_dir> <write_dir> <chunks> <out.json> <label> <kernel source files...>
        HBM bytes that cross the interface per mrp_phase_reads_many call over <chunks> chunks (tools/pipeline_probe.py, ONE batch):
        sum over EVERY dispatch of every mrp_ kernel of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (separate --pmc passes; gfx950 tallies a
        wide coalesced read at one half, MI355X_MICROARCH.md "HBM"), divided by the number of calls in the trace (one
        mrp_traceback_kernel per call).  Per kernel family the same sum is listed.  The file is stamped with the SHA-256 of the kernel
        sources: bench.py quotes it only while they are unchanged.
  r05_summary.py stats <dir>   (as r04_summary.py stats)"""
import collections, csv, glob, hashlib, json, sys

import r04_summary


def rows_of(d, pattern):
    out = []
    for f in glob.glob(d + "/**/" + pattern, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def short(n):
    return n.split("(")[0].split("<")[0].replace("void ", "")


def family_sums(d, counter):
    by, calls = collections.Counter(), 0
    for r in rows_of(d, "*counter_collection.csv"):
        if "mrp_" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        by[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    for r in rows_of(d, "*counter_collection.csv"):
        if "mrp_traceback_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            calls += 1
    return by, calls


def path_traffic(fd, wd, chunks, out, where, *kernel_files):
    f, fc = family_sums(fd, "FETCH_SIZE")
    w, wc = family_sums(wd, "WRITE_SIZE")
    assert fc > 0 and wc > 0, "no mrp_traceback_kernel dispatch in the trace"
    fam = {}
    for k in sorted(set(f) | set(w)):
        fam[k] = dict(fetch_bytes_2x=2.0 * 1024.0 * f.get(k, 0.0) / fc, write_bytes=1024.0 * w.get(k, 0.0) / wc)
        fam[k]["hbm_bytes"] = fam[k]["fetch_bytes_2x"] + fam[k]["write_bytes"]
    tot = sum(v["hbm_bytes"] for v in fam.values())
    h = hashlib.sha256()
    for kf in kernel_files:
        h.update(open(kf, "rb").read())
    json.dump(dict(chunks=int(chunks), hbm_bytes_per_batch=tot, calls_in_fetch_pass=fc, calls_in_write_pass=wc, kernel_sources_sha256=h.hexdigest(),
                   source=where + ": rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over every kernel of mrp_phase_reads_many on "
                          f"{chunks} configs[1] chunks in ONE batch (tools/pipeline_probe.py); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over all dispatches "
                          "of a call; bench.py scales it to the step's chunks",
                   families=dict(sorted(fam.items(), key=lambda kv: -kv[1]["hbm_bytes"]))), open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    if sys.argv[1] == "path_traffic":
        path_traffic(*sys.argv[2:])
    else:
        r04_summary.stats(sys.argv[2])
