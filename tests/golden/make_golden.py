"""Generates tests/golden/*.npz.

PROVENANCE: these vectors are produced by the repo's CPU oracle (oracle/rphmm_oracle.c), NOT by the
reference: the reference cannot be built in this environment (sonLib/htslib submodules are empty)
and its own hot-path tests are unseeded, so it holds no forward/backward golden vectors.  They
freeze the oracle's outputs (which the CPU suite cross-checks against an independent brute-force
evaluation and the reference's invariants) so that a later change to oracle OR kernels that alters
any value is caught.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from margin_amd import synth  # noqa: E402
from oracle import orc        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ["col_ref_start", "col_length", "col_depth", "col_cell_off", "col_read_off", "read_byte_off", "partition",
        "mask_from", "mask_to", "mcol_cell_off", "merge_from", "merge_to", "cell_next", "cell_prev", "cell_forward",
        "cell_backward", "merge_forward", "merge_backward", "col_total"]

CASES = {
    # name: (chunk factory, params, how many of the largest jobs to keep)
    "ont_max_mode": (lambda: synth.make_ont_chunk(seed=41, region_bp=15_000, n_sites=30, coverage=16),
                     dict(synth.shipped_phase_params(), maxPartitionsInAColumn=16, minPartitionsInAColumn=16), 6),
    "unit_sum_mode": (lambda: synth.make_unit_test_chunk(seed=42, ref_length=30, coverage=8, min_read=5, max_read=20,
                                                         error_rate=0.05),
                      dict(synth.unit_test_params(max_partitions=12, max_not_sum=0), includeAncestorSubProb=1), 6),
}


def main():
    for name, (factory, pd, keep) in CASES.items():
        chunk = factory()
        oc = orc.OracleChunk(chunk)
        res = oc.phase(pd, capture_jobs=True)
        oc.close()
        jobs = sorted(res["jobs"], key=lambda j: -len(j["partition"]))[:keep]
        out = dict(allele_number=chunk.allele_number, sub=chunk.sub, prior=chunk.prior, pool=chunk.pool,
                   n_jobs=np.int64(len(jobs)), hap1=res["hap1"], hap2=res["hap2"],
                   reads1=np.array(sorted(res["reads1"]), dtype=np.int64),
                   reads2=np.array(sorted(res["reads2"]), dtype=np.int64))
        for i, j in enumerate(jobs):
            for k in KEYS:
                out[f"j{i}_{k}"] = np.asarray(j[k])
            out[f"j{i}_scalars"] = np.array([j["n_columns"], j["flags"]], dtype=np.int64)
            out[f"j{i}_hmm_fb"] = np.array([j["hmm_forward"], j["hmm_backward"]], dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, len(jobs), "jobs", sum(len(j["partition"]) for j in jobs), "cells")


PHASE_CASES = {
    # whole phasing runs (bubbleGraph.c:2673 driver): inputs incl. the read table, and every output of the driver
    "phase_ont": (lambda: synth.make_ont_chunk(seed=43, region_bp=40_000, n_sites=80, coverage=24), synth.shipped_phase_params()),
    "phase_unit_max": (lambda: synth.make_unit_test_chunk(seed=44, ref_length=60, coverage=10, min_read=8, max_read=30, error_rate=0.05),
                       dict(synth.unit_test_params(max_partitions=40, max_not_sum=1), includeAncestorSubProb=1,
                            roundsOfIterativeRefinement=2)),
}
PHASE_OUT = ["hap1", "hap2", "genotype", "ancestor", "genotype_probs", "hap_probs1", "hap_probs2", "support1", "support2"]


def main_phase():
    import json
    for name, (factory, pd) in PHASE_CASES.items():
        chunk = factory()
        oc = orc.OracleChunk(chunk)
        res = oc.phase(pd)
        oc.close()
        out = dict(allele_number=chunk.allele_number, sub=chunk.sub, prior=chunk.prior, pool=chunk.pool,
                   read_ref_start=np.array([r.ref_start for r in chunk.reads], dtype=np.int32),
                   read_length=np.array([r.length for r in chunk.reads], dtype=np.int32),
                   read_strand=np.array([r.strand for r in chunk.reads], dtype=np.int32),
                   read_pool_off=np.array([r.pool_off for r in chunk.reads], dtype=np.int64),
                   read_names=np.array([r.name for r in chunk.reads]), params=np.array(json.dumps(pd)),
                   ref_start=np.int64(res["ref_start"]), length=np.int64(res["length"]),
                   reads1=np.array(res["reads1"], dtype=np.int64), reads2=np.array(res["reads2"], dtype=np.int64),
                   fb_calls=np.int64(res["fb_calls"]))
        for k in PHASE_OUT:
            out[k] = np.asarray(res[k])
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, len(chunk.reads), "reads", int(res["length"]), "sites", int(res["fb_calls"]), "sweeps")


if __name__ == "__main__":
    main()
    main_phase()
