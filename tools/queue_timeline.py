#!/usr/bin/env python3
"""One resident call and one queue call of the same chunks with MRP_TIMING on (run on the GPU box; stderr carries the timeline)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from margin_amd import capi, sharding, synth
ap = argparse.ArgumentParser(); ap.add_argument("--chunks", type=int, default=576); ap.add_argument("--only", default="", help="resident | queue: just that leg, four calls (for a kernel trace)"); a = ap.parse_args()
params = capi.Params.from_reference_names(synth.shipped_phase_params())
capi.load().mrp_set_host_threads(min(16, os.cpu_count() or 8))
with ThreadPoolExecutor(max_workers=16) as ex:
    chunks = list(ex.map(lambda s: synth.make_ont_chunk(seed=s, region_bp=2000 * 500, n_sites=2000, coverage=30), sharding.chunk_seeds(0, a.chunks)))
for c in chunks:
    capi.read_records(c)
if a.only == "queue":
    descs = capi.chunk_descs(chunks)
    q = capi.Queue([0])
    for _ in range(4):
        q.phase(chunks, params, chunks_per_batch=0, descs=descs, convert=False)
    q.close()
    sys.exit(0)
ctx = capi.Context(0)
dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
if a.only == "resident":
    for _ in range(4):
        capi.phase_reads_many(ctx, dch, chunks, params, convert=False)
    sys.exit(0)
os.environ.pop("MRP_TIMING", None)
for _ in range(3):
    capi.phase_reads_many(ctx, dch, chunks, params, convert=False)
descs = capi.chunk_descs(chunks)
q = capi.Queue([0])
for _ in range(3):
    q.phase(chunks, params, chunks_per_batch=0, descs=descs, convert=False)
os.environ["MRP_TIMING"] = "1"
print("==== resident", file=sys.stderr, flush=True)
t0 = time.perf_counter(); capi.phase_reads_many(ctx, dch, chunks, params, convert=False); print(f"==== resident took {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
print("==== queue", file=sys.stderr, flush=True)
t0 = time.perf_counter(); q.phase(chunks, params, chunks_per_batch=0, descs=descs, convert=False); print(f"==== queue took {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
q.close()
