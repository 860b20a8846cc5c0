#!/usr/bin/env python3
"""Development probe: the headline step (one mrp_phase_reads_many call over N configs[1] chunks, inputs resident) under a list of
settings that can be changed at run time, chunks synthesised ONCE.  usage: step_probe.py [--chunks 1152] [--steps 6] SETTING ...
  SETTING = comma-separated KEY=VALUE: threads=<host pool threads>, groups=<concurrent batches>, or any environment variable the library
  reads per call (MRP_...).  Prints the median and all step times per setting."""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from margin_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=1152)
    ap.add_argument("--sites", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("settings", nargs="*", default=["threads=16"])
    args = ap.parse_args()
    params = capi.Params.from_reference_names(synth.shipped_phase_params())
    with ThreadPoolExecutor(max_workers=16) as ex:
        chunks = list(ex.map(lambda s: synth.make_ont_chunk(seed=s + 1, region_bp=args.sites * 500, n_sites=args.sites, coverage=30.0), range(args.chunks)))
    units = sum(c.units for c in chunks)
    for c in chunks:
        capi.read_records(c)
    capi.load().mrp_set_host_threads(16)
    ctx = capi.Context(0)
    dchunks = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
    prepared = capi.phase_many_args(dchunks, chunks)
    for _ in range(2):
        capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=prepared)
    for setting in args.settings:
        saved = {}
        for kv in setting.split(","):
            k, v = kv.split("=", 1)
            if k == "threads":
                capi.load().mrp_set_host_threads(int(v))
            elif k == "groups":
                ctx.set_phase_groups(int(v))
            else:
                saved[k] = os.environ.get(k)
                os.environ[k] = v
        capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=prepared)
        ms, cpu0 = [], time.process_time()
        for _ in range(args.steps):
            t0 = time.perf_counter()
            _, st = capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=prepared)
            ms.append(1e3 * (time.perf_counter() - t0))
        cpu = (time.process_time() - cpu0) / args.steps
        med = sorted(ms)[len(ms) // 2]
        print(f"{setting:40s} median {med:7.1f} ms = {units / med / 1e3:.3e} units/s  host cpu {cpu:.2f} s  runs {[round(x, 1) for x in ms]}  "
              f"(pack {st.pack_ms:.0f} xe {st.cross_emit_ms:.0f} rec {st.recursion_ms:.0f} prune {st.prune_kernel_ms:.0f} compact {st.compact_ms:.0f})", flush=True)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        capi.load().mrp_set_host_threads(16)
        ctx.set_phase_groups(0)


if __name__ == "__main__":
    main()
