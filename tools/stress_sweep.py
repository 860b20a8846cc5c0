"""Developer stress tool (test infrastructure): several host threads sweep the same job set
concurrently; every result is compared bit for bit with the oracle and mismatches are located."""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from margin_amd import capi, synth
from oracle import orc

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
chunk = synth.make_ont_chunk(seed=11, region_bp=400_000, n_sites=800, coverage=30)
oc = orc.OracleChunk(chunk)
flats = oc.phase(synth.shipped_phase_params(), capture_jobs=True)["jobs"]
if len(sys.argv) > 3:
    lim = int(sys.argv[3])
    flats = [f for f in flats if int(np.diff(f["col_cell_off"]).max()) <= lim]
print("jobs", len(flats), "cells", sum(len(f["partition"]) for f in flats), flush=True)
bad = []
def worker(w):
    ctx = capi.Context(0)
    dchunk = capi.DeviceChunk.from_chunk(ctx, chunk)
    for it in range(iters):
        jobs = [capi.Job(dchunk, f, int(f["flags"])) for f in flats]
        capi.fb_run(ctx, jobs)
        for hi, (f, j) in enumerate(zip(flats, jobs)):
            r = j.results()
            for name in ("cell_forward", "cell_backward", "merge_forward", "merge_backward", "col_total"):
                x, y = np.asarray(f[name]), np.asarray(r[name])
                neq = ~((x == y) | (np.isneginf(x) & np.isneginf(y)))
                if neq.any():
                    idx = np.nonzero(neq)[0]
                    off = f["col_cell_off"] if name.startswith("cell") else f["mcol_cell_off"]
                    cols = np.searchsorted(off, idx, side="right") - 1
                    bad.append((w, it, hi, name, len(idx), int(idx[0]), int(idx[-1]), int(cols[0]), int(cols[-1]), f["n_columns"], len(f["partition"]), int(np.diff(f["col_cell_off"]).max()), x[idx[:4]].tolist(), y[idx[:4]].tolist()))
    dchunk.close(); ctx.close()
ts = [threading.Thread(target=worker, args=(w,)) for w in range(n_threads)]
t0 = time.time()
[t.start() for t in ts]; [t.join() for t in ts]
print("done", time.time() - t0, "s; mismatching (thread, iter, hmm, array, n_bad, first, last, firstcol, lastcol, K, cells, maxcells, want, got):")
for b in bad[:30]: print(b)
print("total mismatches", len(bad))
