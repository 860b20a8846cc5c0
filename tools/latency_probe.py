#!/usr/bin/env python3
"""Development probe: end-to-end latency of mrp_phase_reads_many over a few chunks (inputs resident): 1 and 8 chunks of 2 000 sites,
1, 8 and 640 chunks of 130 sites.  usage: latency_probe.py [reps]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from margin_amd import capi, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
params = capi.Params.from_reference_names(synth.shipped_phase_params())
capi.load().mrp_set_host_threads(32)
ctx = capi.Context(0)
with ThreadPoolExecutor(max_workers=16) as ex:
    big = list(ex.map(lambda s: synth.make_ont_chunk(seed=s + 1, region_bp=2000 * 500, n_sites=2000, coverage=30.0), range(8)))
    small = list(ex.map(lambda s: synth.make_ont_chunk(seed=50_000 + s, region_bp=130 * 500, n_sites=130, coverage=30.0), range(640)))
for name, cs in (("1 x 2000 sites", big[:1]), ("8 x 2000 sites", big), ("1 x 130 sites", small[:1]), ("8 x 130 sites", small[:8]), ("640 x 130 sites", small)):
    for c in cs:
        capi.read_records(c)
    d = [capi.DeviceChunk.from_chunk(ctx, c) for c in cs]
    prep = capi.phase_many_args(d, cs)
    for _ in range(2):
        capi.phase_reads_many(ctx, d, cs, params, convert=False, prepared=prep)
    ms = []
    for _ in range(reps):
        t0 = time.perf_counter()
        capi.phase_reads_many(ctx, d, cs, params, convert=False, prepared=prep)
        ms.append(1e3 * (time.perf_counter() - t0))
    u = sum(c.units for c in cs)
    med = sorted(ms)[len(ms) // 2]
    print(f"{name:18s} median {med:7.2f} ms  min {min(ms):7.2f}  {u / med / 1e3:.3e} units/s", flush=True)
    for x in d:
        x.close()
