#!/usr/bin/env python3
"""What a maintainer gets who links ONLY the adaptor (integration/stRPHmm_forwardBackward_adaptor.c) into margin and leaves
everything else on the CPU: the oracle's phasing driver (CPU restatement of bubbleGraph_phaseBubbleGraph: tiling paths, cross
products by hashing, prune, trace back) with the product's adaptor in the seam --

    oracle        its own stRPHmm_forwardBackward on the CPU (the reference's path)
    per_hmm       stRPHmm_forwardBackward (hmm.c:931) replaced: one flatten + one mrp_fb_run per hmm, ~420 per chunk
    per_merge     the sweep loop of one mergeTwoTilingPaths call (coordination.c:285-328) replaced by
                  stRPHmm_forwardBackwardMany: the independent cross products of a call in one device batch

-- one chunk per host thread (phase.c:276), beside the whole-chunk resident path (mrp_phase_reads_many) on the same chunks.
Test infrastructure (it drives the oracle); the numbers it prints are the ones quoted in INTEGRATION.md.
usage: adaptor_probe.py [--chunks 4] [--threads 4] [--sites 2000]"""
import argparse
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from margin_amd import capi, synth  # noqa: E402
from oracle import orc  # noqa: E402
from tests.test_adaptor import build_adaptor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--sites", type=int, default=2000)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--min-cells", type=int, nargs="+", default=[0, 1024, 4096, 16384, 65536])
    args = ap.parse_args()
    L = orc.lib()
    A = build_adaptor(orc)
    L.orc_set_fb_override.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_set_fb_override.restype = None
    one = C.cast(A.stRPHmm_forwardBackward, C.c_void_p)
    many = C.cast(A.stRPHmm_forwardBackwardMany, C.c_void_p)
    pd = synth.shipped_phase_params()
    chunks = [synth.make_ont_chunk(seed=s + 1, region_bp=args.sites * 500, n_sites=args.sites, coverage=args.coverage) for s in range(args.chunks)]
    units = float(sum(c.units for c in chunks))
    ocs = [orc.OracleChunk(c) for c in chunks]
    for oc in ocs:  # the adaptor's fast path: the reads' profile bytes of a chunk uploaded once
        A.adp_test_register(oc.ref, oc.seq_array(), len(oc.seqs))

    def run(label, o, m):
        L.orc_set_fb_override(o, m)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=args.threads) as ex:
            res = list(ex.map(lambda oc: oc.phase(pd), ocs))
        dt = time.perf_counter() - t0
        L.orc_set_fb_override(None, None)
        fb = sum(r["fb_seconds"] for r in res)
        calls = sum(r["fb_calls"] for r in res)
        print(f"{label:20s}: {dt:7.2f} s wall for {len(ocs)} chunks on {args.threads} threads = {units / dt:10.3e} units/s; inside the seam "
              f"{fb:7.2f} thread-s ({calls} sweeps), outside {dt * min(args.threads, len(ocs)) - fb:7.2f} thread-s", flush=True)
        return res, dt

    ref, t_ref = run("oracle", None, None)
    a = b = None
    for thr in args.min_cells:  # hmms with fewer cells than this stay on the CPU body (mrpAdaptor_setMinCells)
        A.adp_test_set_min_cells(thr)
        a, t_a = run(f"per_hmm   >= {thr}", one, None)
        b, t_b = run(f"per_merge >= {thr}", one, many)
    for r0, r1, r2 in zip(ref, a, b):
        for k in ("hap1", "hap2", "genotype"):
            assert (r0[k] == r1[k]).all() and (r0[k] == r2[k]).all(), k
        assert r0["reads1"] == r1["reads1"] == r2["reads1"]
    print("haplotypes and read partitions identical in all three runs")
    for oc in ocs:
        A.adp_test_unregister(oc.ref)
    A.adp_test_cleanup()
    # the whole-chunk path on the same chunks
    ctx = capi.Context(0)
    params = capi.Params.from_reference_names(pd)
    dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
    for c in chunks:
        capi.read_records(c)
    capi.phase_reads_many(ctx, dch, chunks, params, convert=False)
    t0 = time.perf_counter()
    got, st = capi.phase_reads_many(ctx, dch, chunks, params)
    t_res = time.perf_counter() - t0
    for r0, g in zip(ref, got):
        assert (np.asarray(g["hap1"]) == r0["hap1"]).all() and g["reads1"] == r0["reads1"]
    print(f"resident  : {t_res:7.3f} s wall for the same {len(chunks)} chunks in one mrp_phase_reads_many call = {units / t_res:10.3e} units/s "
          f"(identical results; {units / t_res / (units / t_ref):.0f}x the oracle on {args.threads} threads)")
    for oc in ocs:
        oc.close()


if __name__ == "__main__":
    main()
