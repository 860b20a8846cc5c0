#!/usr/bin/env python3
"""Timing probe of mrp_forward_probabilities on the alignment pairs of config-2 chunks (2 000 sites x ~30 reads x 2 alleles
per chunk): kernel time (HIP events), call time, cell updates per second, optional oracle check and CPU timing."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from margin_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--sites", type=int, default=2000)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--check", type=int, default=2000, help="pairs compared with the oracle")
    args = ap.parse_args()
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    f = capi.PairHmm.from_margin_hmm(t, tr, em)
    models = [f, f.reverse_complement()]
    t0 = time.time()
    bubbles = []
    for c in range(args.chunks):
        bubbles += synth.make_bubble_strings(seed=c + 1, n_sites=args.sites, coverage=args.coverage)
    pool, xo, xl, yo, yl, mi = synth.pairs_from_bubbles(bubbles)
    print(f"synth {time.time() - t0:.1f}s: {len(xo)} pairs, pool {pool.size} B", flush=True)
    ctx = capi.Context(0)
    for r in range(args.repeat):
        out, st = capi.forward_probabilities(ctx, models, pool, xo, xl, yo, yl, mi)
        print(f"run {r}: kernel {st.kernel_ms:.3f} ms, call {st.total_ms:.2f} ms, {len(xo) / st.kernel_ms * 1e3:.3e} pairs/s, "
              f"{st.cells / st.kernel_ms * 1e3:.3e} cells/s (kernel), {len(xo) / st.total_ms * 1e3:.3e} pairs/s (call); lane {st.pairs_lane} wave {st.pairs_wave}",
              flush=True)
    if args.check:
        from oracle import pairhmm as ph
        rng = np.random.default_rng(0)
        pick = np.sort(rng.choice(len(out), size=min(args.check, len(out)), replace=False))
        om = [ph.Model.from_buffer_copy(bytes(m)) for m in models]
        t0 = time.perf_counter()
        ref = ph.forward_batch(om, pool, xo[pick], xl[pick], yo[pick], yl[pick], mi[pick])
        dt = time.perf_counter() - t0
        print(f"oracle: {len(pick) / dt:.3e} pairs/s on one core; identical={bool((ref == out[pick]).all())}", flush=True)


if __name__ == "__main__":
    main()
