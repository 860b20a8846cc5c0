#!/usr/bin/env python3
"""End-to-end probe of the device-resident phasing pipeline (mrp_phase_reads_many) on config-2 chunks:
wall time, engine statistics, and optional parity against the per-chunk host path / the oracle."""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from margin_amd import capi, synth  # noqa: E402

KEYS = ("hap1", "hap2", "genotype", "ancestor", "support1", "support2", "genotype_probs", "hap_probs1", "hap_probs2")


def same(a, b):
    return all((np.asarray(a[k]) == np.asarray(b[k])).all() for k in KEYS) and a["reads1"] == b["reads1"] and a["reads2"] == b["reads2"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--sites", type=int, default=2000)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--check-host", type=int, default=1, help="compare this many chunks with mrp_phase_reads")
    ap.add_argument("--check-oracle", type=int, default=0)
    ap.add_argument("--groups", type=int, default=1, help="phase the chunks as this many concurrent batches (one host thread + context each)")
    ap.add_argument("--hifi", action="store_true", help="HiFi-like chunks (SURVEY.md 8d configs 3-5): 35x, reads N(18 kb, 3 kb), 1 %% allele error, 2-4 alleles per site")
    ap.add_argument("--sample", default="", help="CPU sampling profile of the timed runs after the first into this file (tools/sampler)")
    args = ap.parse_args()
    pd = synth.shipped_phase_params()
    params = capi.Params.from_reference_names(pd)
    if os.environ.get("MRP_PROBE_BIND"):  # the CPUs next to the device, as bench.py's ranks do
        import bench
        capi.load().mrp_device_count()
        print("cpu binding:", bench.bind_near_device(0), flush=True)
    ctx = capi.Context(0)
    ctx.set_test_hooks(int(os.environ.get("MRP_TEST_HOOKS", "0")))  # bit 1: separate cross product / emission kernels (A/B)
    if os.environ.get("MRP_PHASE_GROUPS"):
        ctx.set_phase_groups(int(os.environ["MRP_PHASE_GROUPS"]))
    if os.environ.get("MRP_HOST_THREADS"):
        capi.load().mrp_set_host_threads(int(os.environ["MRP_HOST_THREADS"]))
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=min(16, args.chunks)) as ex:
        if args.hifi:
            make = lambda s: synth.make_ont_chunk(seed=s + 1, region_bp=args.sites * 500, n_sites=args.sites, coverage=35.0, median_len=18_000.0,
                                                  allele_error=0.01, allele_choices=(2, 3, 4), allele_probs=(0.85, 0.1, 0.05), length_model="normal",
                                                  normal_sd=3000.0)
        else:
            make = lambda s: synth.make_ont_chunk(seed=s + 1, region_bp=args.sites * 500, n_sites=args.sites, coverage=args.coverage)
        chunks = list(ex.map(make, range(args.chunks)))
    print(f"synth {time.time() - t0:.2f}s", flush=True)
    dchunks = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
    units = sum(c.units for c in chunks)
    for c in chunks:
        capi.read_records(c)
    if args.groups > 1:
        G = args.groups
        gctx = [capi.Context(0) for _ in range(G)]
        gd = [[capi.DeviceChunk.from_chunk(gctx[g], c) for c in chunks[g::G]] for g in range(G)]
        for r in range(args.repeat):
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=G) as ex:
                res = list(ex.map(lambda g: capi.phase_reads_many(gctx[g], gd[g], chunks[g::G], params, convert=False), range(G)))
            dt = time.perf_counter() - t0
            print(f"run {r} ({G} concurrent batches): {dt * 1e3:.1f} ms wall, {units / dt:.3e} units/s, device_ms per batch "
                  f"{[round(x[1].device_ms, 1) for x in res]}", flush=True)
    sampler = None
    for r in range(args.repeat):
        last = r + 1 == args.repeat
        if args.sample and r == 1:
            import ctypes
            sampler = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sampler", "libcpusampler.so"))
            sampler.cpusampler_start(int(os.environ.get("MRP_SAMPLE_HZ", "250")))
        t0, c0 = time.perf_counter(), time.process_time()
        got, st = capi.phase_reads_many(ctx, dchunks, chunks, params, convert=last)
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        print(f"run {r}: {dt * 1e3:.1f} ms wall, {units / dt:.3e} units/s, host cpu {cpu * 1e3:.0f} ms, resident={st.resident} levels={st.levels} hmms={st.hmms} "
              f"cols={st.columns} cells={st.cells} device_ms={st.device_ms:.2f} (cross {st.cross_ms:.2f} sweep {st.sweep_ms:.2f} "
              f"prune {st.prune_ms:.2f}; pack {st.pack_ms:.2f} cross+emission {st.cross_emit_ms:.2f} recursion {st.recursion_ms:.2f} prune kernels {st.prune_kernel_ms:.2f} "
              f"compaction {st.compact_ms:.2f})", flush=True)
    if sampler is not None:
        sampler.cpusampler_stop(args.sample.encode())
    for i in range(min(args.check_host, args.chunks)):
        t0 = time.perf_counter()
        host = capi.phase_reads(ctx, dchunks[i], chunks[i], params)
        print(f"chunk {i}: host path {time.perf_counter() - t0:.2f}s, identical={same(got[i], host)}", flush=True)
    if args.check_oracle:
        from oracle import orc
        for i in range(min(args.check_oracle, args.chunks)):
            oc = orc.OracleChunk(chunks[i])
            t0 = time.perf_counter()
            ref = oc.phase(pd)
            oc.close()
            print(f"chunk {i}: oracle {time.perf_counter() - t0:.2f}s, identical={same(got[i], ref)}", flush=True)


if __name__ == "__main__":
    main()
