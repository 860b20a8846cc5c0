#!/usr/bin/env python3
"""Development: is the SECOND family of contexts of a process slower than the first?  (queue then resident call then queue ...)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from margin_amd import capi, sharding, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 576
first = sys.argv[2] if len(sys.argv) > 2 else "queue"
params = capi.Params.from_reference_names(synth.shipped_phase_params())
capi.load().mrp_set_host_threads(16)
with ThreadPoolExecutor(max_workers=16) as ex:
    chunks = list(ex.map(lambda s: synth.make_ont_chunk(seed=s, region_bp=2000 * 500, n_sites=2000, coverage=30), sharding.chunk_seeds(0, N)))
for c in chunks:
    capi.read_records(c)
descs = capi.chunk_descs(chunks)
state = {}
def queue_runs(tag):
    if "q" not in state: state["q"] = capi.Queue([0])
    t = []
    for r in range(4):
        t0 = time.perf_counter(); state["q"].phase(chunks, params, chunks_per_batch=0, descs=descs, convert=False); t.append(1e3 * (time.perf_counter() - t0))
    print(f"{tag:34s} queue    " + " ".join(f"{x:7.1f}" for x in t), flush=True)
def resident_runs(tag):
    if "ctx" not in state:
        state["ctx"] = capi.Context(0); state["d"] = [capi.DeviceChunk.from_chunk(state["ctx"], c) for c in chunks]
        state["prep"] = capi.phase_many_args(state["d"], chunks)
    t = []
    for r in range(4):
        t0 = time.perf_counter(); capi.phase_reads_many(state["ctx"], state["d"], chunks, params, convert=False, prepared=state["prep"]); t.append(1e3 * (time.perf_counter() - t0))
    print(f"{tag:34s} resident " + " ".join(f"{x:7.1f}" for x in t), flush=True)
order = [queue_runs, resident_runs] if first == "queue" else [resident_runs, queue_runs]
order[0]("first family"); order[1]("second family"); order[0]("first family again"); order[1]("second family again")
