#!/usr/bin/env python3
"""Where the work queue's time goes against a resident call of the same chunks (run on the GPU box):
   resident natural order / resident in the queue's cost order / queue / queue with the first context gone."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from margin_amd import capi, sharding, synth

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=576)
ap.add_argument("--runs", type=int, default=5)
a = ap.parse_args()
params = capi.Params.from_reference_names(synth.shipped_phase_params())
capi.load().mrp_set_host_threads(min(16, os.cpu_count() or 8))
seeds = sharding.chunk_seeds(0, a.chunks + 64)
with ThreadPoolExecutor(max_workers=16) as ex:
    chunks = list(ex.map(lambda s: synth.make_ont_chunk(seed=s, region_bp=2000 * 500, n_sites=2000, coverage=30), seeds))
for c in chunks:
    capi.read_records(c)

def timed(name, f):
    f()
    t = []
    for _ in range(a.runs):
        t0 = time.perf_counter(); f(); t.append(1e3 * (time.perf_counter() - t0))
    print(f"{name:44s} min {min(t):7.1f}  median {sorted(t)[len(t)//2]:7.1f}  max {max(t):7.1f} ms", flush=True)

ctx = capi.Context(0)
all_chunks = chunks
all_dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in all_chunks]
chunks, dch = all_chunks[:a.chunks], all_dch[:a.chunks]
timed("resident, natural order", lambda: capi.phase_reads_many(ctx, dch, chunks, params, convert=False))
cost = np.array([sum(len(r.sites) if hasattr(r, "sites") else 0 for r in c.reads) if hasattr(c, "reads") else c.units for c in chunks])
order = np.argsort(-np.array([c.units for c in chunks]), kind="stable")
dch2 = [dch[i] for i in order]; ch2 = [chunks[i] for i in order]
timed("resident, cost order", lambda: capi.phase_reads_many(ctx, dch2, ch2, params, convert=False))
k = [0]
def rotating():
    k[0] = (k[0] + 13) % 64
    capi.phase_reads_many(ctx, all_dch[k[0]:k[0] + a.chunks], all_chunks[k[0]:k[0] + a.chunks], params, convert=False)
timed("resident, a different window of chunks per call", rotating)
timed("resident, natural order again", lambda: capi.phase_reads_many(ctx, dch, chunks, params, convert=False))
descs = capi.chunk_descs(chunks)
q = capi.Queue([0])
timed("queue (first context alive, untrimmed)", lambda: q.phase(chunks, params, chunks_per_batch=0, descs=descs, convert=False))
ctx.trim()
timed("queue (first context trimmed)", lambda: q.phase(chunks, params, chunks_per_batch=0, descs=descs, convert=False))
timed("resident again", lambda: capi.phase_reads_many(ctx, dch, chunks, params, convert=False))
q.close()
