#!/usr/bin/env python3
"""Development probe: one GPU's eighth of a whole genome (BASELINE.json configs[3] / 8: N chunks of ~130 het sites) from HOST memory through the
work queue on device 0, as bench.py's configs[3]/8 leg runs it, and the same chunks as ONE resident mrp_phase_reads_many call.
usage: genome_probe.py [--chunks 3900] [--runs 5] [--threads 32]"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from margin_amd import capi, synth  # noqa: E402


def cpu_stat():
    try:
        return {k: int(v) for k, v in (line.split() for line in open("/sys/fs/cgroup/cpu.stat"))}
    except (OSError, ValueError):
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=3900)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--resident", type=int, default=1)
    args = ap.parse_args()
    params = capi.Params.from_reference_names(synth.shipped_phase_params())
    with ThreadPoolExecutor(max_workers=16) as ex:
        gcs = list(ex.map(lambda s: synth.make_ont_chunk(seed=50_000 + s, region_bp=130 * 500, n_sites=130, coverage=30.0), range(args.chunks)))
    for c in gcs:
        capi.read_records(c)
    units = float(sum(c.units for c in gcs))
    capi.load().mrp_set_host_threads(args.threads)
    if args.resident:
        ctx = capi.Context(0)
        gd = [capi.DeviceChunk.from_chunk(ctx, c) for c in gcs]
        prepared = capi.phase_many_args(gd, gcs)
        for _ in range(2):
            capi.phase_reads_many(ctx, gd, gcs, params, convert=False, prepared=prepared)
        ms, c0 = [], time.process_time()
        for _ in range(args.runs):
            t0 = time.perf_counter()
            capi.phase_reads_many(ctx, gd, gcs, params, convert=False, prepared=prepared)
            ms.append(1e3 * (time.perf_counter() - t0))
        med = sorted(ms)[len(ms) // 2]
        print(f"resident: {args.chunks} chunks, median {med:.1f} ms = {units / med / 1e3:.3e} units/s, host cpu {(time.process_time() - c0) / args.runs:.2f} s per call, runs {[round(x, 1) for x in ms]}", flush=True)
        for d in gd:
            d.close()
        ctx.trim()
    q = capi.Queue([0])
    descs = capi.chunk_descs(gcs)
    for _ in range(2):
        q.phase(gcs, params, chunks_per_batch=0, descs=descs, convert=False)
    ms, c0, th0 = [], time.process_time(), cpu_stat()
    for _ in range(args.runs):
        t0 = time.perf_counter()
        _, st = q.phase(gcs, params, chunks_per_batch=0, descs=descs, convert=False)
        ms.append(1e3 * (time.perf_counter() - t0))
    med = sorted(ms)[len(ms) // 2]
    th1 = cpu_stat()
    print(f"queue:    {args.chunks} chunks in {int(st.batches)} batches, median {med:.1f} ms = {units / med / 1e3:.3e} units/s, host cpu {(time.process_time() - c0) / args.runs:.2f} s per run, "
          f"runs {[round(x, 1) for x in ms]}, cgroup: {th1.get('nr_throttled', 0) - th0.get('nr_throttled', 0)} throttled periods, "
          f"{(th1.get('throttled_usec', 0) - th0.get('throttled_usec', 0)) / 1e3:.0f} ms", flush=True)
    q.close()


if __name__ == "__main__":
    main()
