"""Developer probe (test infrastructure): time the HIP sweep on oracle-generated jobs of one
config-2 chunk, replicated R times in one device batch.  Not the benchmark (bench.py builds its
jobs with the product host pipeline)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from margin_amd import capi, synth
from oracle import orc

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
region = n_sites * 500
chunk = synth.make_ont_chunk(seed=1, region_bp=region, n_sites=n_sites, coverage=30)
oc = orc.OracleChunk(chunk)
t = time.time()
res = oc.phase(synth.shipped_phase_params(), capture_jobs=True)
print(f"oracle: {time.time()-t:.1f}s total, fb {res['fb_seconds']:.1f}s in {res['fb_calls']} sweeps; units {chunk.units}", flush=True)
flats = res["jobs"]
cells = sum(len(f["partition"]) for f in flats)
print("jobs", len(flats), "cells", cells, "max cells/col", max(int(np.diff(f["col_cell_off"]).max()) for f in flats), flush=True)
ctx = capi.Context(0)
dchunk = capi.DeviceChunk.from_chunk(ctx, chunk)
b = capi.Batch(ctx)
t = time.time()
for r in range(R):
    for f in flats:
        b.add(capi.Job(dchunk, f, int(f["flags"])), keep=(r == 0))
print(f"add {time.time()-t:.1f}s", flush=True)
t = time.time(); b.upload(); print(f"upload {time.time()-t:.1f}s", flush=True)
for it in range(3):
    b.launch()
    s = b.stats()
    tot_ms = s.planes_ms + s.emission_ms + s.sweep_ms; gbs = s.algorithmic_bytes / (tot_ms * 1e-3) / 1e9
    print(f"iter {it}: planes {s.planes_ms:.3f} ms emis {s.emission_ms:.3f} ms sweep {s.sweep_ms:.3f} ms  cells {s.n_cells}  alg {s.algorithmic_bytes/1e9:.2f} GB -> {gbs:.1f} GB/s  units/s {R*chunk.units/(tot_ms*1e-3):.3e}", flush=True)
b.download()
for f, j in zip(flats, b.jobs):
    r = j.results()
    assert (r["cell_forward"] == f["cell_forward"]).all() and (r["cell_backward"] == f["cell_backward"]).all()
    assert (r["col_total"] == f["col_total"]).all()
print("parity ok (replica 0)")
print(f"cpu oracle fb: {chunk.units/res['fb_seconds']:.3e} units/s (1 thread)")
