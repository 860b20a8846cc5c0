import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from concurrent.futures import ThreadPoolExecutor
from margin_amd import capi, sharding, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1152
params = capi.Params.from_reference_names(synth.shipped_phase_params())
capi.load().mrp_set_host_threads(16)
seeds = sharding.chunk_seeds(0, N)
with ThreadPoolExecutor(max_workers=16) as ex:
    chunks = list(ex.map(lambda s: synth.make_ont_chunk(seed=s, region_bp=2000 * 500, n_sites=2000, coverage=30), seeds))
for c in chunks:
    capi.read_records(c)
descs = capi.chunk_descs(chunks)
q = capi.Queue([0])
for cpb in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]:
    for r in range(4):
        if r == 3 and os.environ.get("QT_TIMING"):
            os.environ["MRP_TIMING"] = "1"
        t0 = time.perf_counter(); c0 = time.process_time()
        _, st = q.phase(chunks, params, chunks_per_batch=cpb, descs=descs, convert=False)
        print(f"queue, {cpb} chunks per batch, run {r}: {1e3*(time.perf_counter()-t0):.1f} ms, cpu {time.process_time()-c0:.2f} s, batches {st.batches}", flush=True)
        os.environ.pop("MRP_TIMING", None)
q.close()
ctx = capi.Context(0)
dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks[:1152]]
for r in range(3):
    t0 = time.perf_counter()
    capi.phase_reads_many(ctx, dch, chunks[:1152], params, convert=False)
    print(f"resident 1152 chunks run {r}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
