#!/usr/bin/env python3
"""A queue longer than one batch (run on the GPU box): N chunks from host memory through mrp_queue_phase_chunks with the
library's batch sizes, against one resident call of 576 of them."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from margin_amd import capi, sharding, synth
ap = argparse.ArgumentParser(); ap.add_argument("--chunks", type=int, default=1728); ap.add_argument("--runs", type=int, default=3); ap.add_argument("--batch", type=int, default=0); ap.add_argument("--sites", type=int, default=2000); ap.add_argument("--base", type=int, default=576); ap.add_argument("--skip-resident", type=int, default=0); a = ap.parse_args()
params = capi.Params.from_reference_names(synth.shipped_phase_params())
capi.load().mrp_set_host_threads(min(32, os.cpu_count() or 8))
with ThreadPoolExecutor(max_workers=16) as ex:
    base = list(ex.map(lambda s: synth.make_ont_chunk(seed=s, region_bp=a.sites * 500, n_sites=a.sites, coverage=30), sharding.chunk_seeds(0, a.base)))
for c in base:
    capi.read_records(c)
chunks = [base[i % a.base] for i in range(a.chunks)]
units = sum(c.units for c in chunks)
descs = capi.chunk_descs(chunks)
q = capi.Queue([0])
q.phase(chunks, params, chunks_per_batch=a.batch, descs=descs, convert=False)
for _ in range(a.runs):
    t0 = time.perf_counter(); _, st = q.phase(chunks, params, chunks_per_batch=a.batch, descs=descs, convert=False); dt = time.perf_counter() - t0
    print(f"queue: {a.chunks} chunks in {int(st.batches)} batches: {1e3 * dt:.1f} ms, {units / dt:.3e} units/s, {1e3 * dt / a.chunks * 576:.1f} ms per 576 chunks, {1e3 * dt:.1f} ms in all", flush=True)
q.close()
if a.skip_resident:
    sys.exit(0)
ctx = capi.Context(0)
dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in base]
capi.phase_reads_many(ctx, dch, base, params, convert=False)
for _ in range(a.runs):
    t0 = time.perf_counter(); capi.phase_reads_many(ctx, dch, base, params, convert=False); dt = time.perf_counter() - t0
    print(f"resident: 576 chunks: {1e3 * dt:.1f} ms, {sum(c.units for c in base) / dt:.3e} units/s", flush=True)
