"""Writes tests/golden/pairhmm_forward.npz: string pairs, models, anchors and the forward log probabilities of the CPU
oracle (oracle/pairhmm_oracle.c).  The reference itself cannot be built here (sonLib is absent), so these vectors pin the
ORACLE's numbers, not the reference's: they guard the oracle and the HIP path against drifting together.
Run from the repo root: python tests/golden/make_pairhmm_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from margin_amd import capi, synth  # noqa: E402
from oracle import pairhmm as ph  # noqa: E402


def main():
    rng = np.random.default_rng(20240)
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    f = capi.PairHmm.from_margin_hmm(t, tr, em)
    models = [f, f.reverse_complement(), capi.PairHmm.default_nucleotide()]
    strings, pos = [], 0

    def add(s):
        nonlocal pos
        strings.append(s)
        o = pos
        pos += len(s)
        return o

    out = {"expansion": np.int64(4)}
    # short: margin phase shapes (alleles of 25, noisy reads), a few Ns, every model
    xo, xl, yo, yl, mi = [], [], [], [], []
    for _ in range(120):
        a = synth.random_sequence(rng, int(rng.integers(0, 40)), n_rate=0.02)
        b = synth.evolve_sequence(rng, a) if rng.random() < 0.9 else synth.random_sequence(rng, int(rng.integers(0, 40)))
        xo.append(add(a)); xl.append(len(a)); yo.append(add(b)); yl.append(len(b)); mi.append(int(rng.integers(0, 3)))
    out.update(short_x_off=np.array(xo, np.int64), short_x_len=np.array(xl, np.int32), short_y_off=np.array(yo, np.int64),
               short_y_len=np.array(yl, np.int32), short_model=np.array(mi, np.uint8), short_anchor_off=np.zeros(len(xo) + 1, np.int64),
               short_anchors=np.zeros((0, 2), np.int64), short_ragged=np.array([0, 0], np.uint8))
    # long: anchored bands and ragged ends
    xo, xl, yo, yl, mi, aoff, anc = [], [], [], [], [], [0], []
    for _ in range(24):
        a = synth.random_sequence(rng, int(rng.integers(100, 400)))
        b = synth.evolve_sequence(rng, a)
        xo.append(add(a)); xl.append(len(a)); yo.append(add(b)); yl.append(len(b)); mi.append(int(rng.integers(0, 3)))
        x, y = -1, -1
        while rng.random() < 0.95:
            x += int(rng.integers(1, 21)); y += int(rng.integers(1, 21))
            if x >= len(a) or y >= len(b):
                break
            anc.append((x, y))
        aoff.append(len(anc))
    out.update(long_x_off=np.array(xo, np.int64), long_x_len=np.array(xl, np.int32), long_y_off=np.array(yo, np.int64),
               long_y_len=np.array(yl, np.int32), long_model=np.array(mi, np.uint8), long_anchor_off=np.array(aoff, np.int64),
               long_anchors=np.array(anc, np.int64).reshape(-1, 2), long_ragged=np.array([1, 1], np.uint8))
    pool = np.concatenate(strings)
    om = [ph.Model.from_buffer_copy(bytes(m)) for m in models]
    for tag in ("short", "long"):
        out[f"{tag}_out"] = ph.forward_batch(om, pool, out[f"{tag}_x_off"], out[f"{tag}_x_len"], out[f"{tag}_y_off"], out[f"{tag}_y_len"],
                                             out[f"{tag}_model"], out[f"{tag}_anchor_off"], out[f"{tag}_anchors"], 4,
                                             bool(out[f"{tag}_ragged"][0]), bool(out[f"{tag}_ragged"][1]))
        assert np.isfinite(out[f"{tag}_out"]).all()
    out["pool"] = pool
    out["models"] = np.stack([np.frombuffer(bytes(m), dtype=np.float64) for m in models])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pairhmm_forward.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
