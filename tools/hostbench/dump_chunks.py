#!/usr/bin/env python3
"""Writes the read tables of synthetic configs[1] chunks for tools/hostbench/hostbench.c (CPU-only profile of the host side of the
resident pipeline).  usage: dump_chunks.py <n_chunks> <out.bin> [--sites N]"""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from margin_amd import synth

n, out = int(sys.argv[1]), sys.argv[2]
sites = int(sys.argv[sys.argv.index("--sites") + 1]) if "--sites" in sys.argv else 2000
with open(out, "wb") as f:
    f.write(struct.pack("<q", n))
    for s in range(n):
        c = synth.make_ont_chunk(seed=s + 1, region_bp=sites * 500, n_sites=sites)
        f.write(struct.pack("<qqq", c.n_sites, len(c.reads), c.pool.shape[0]))
        f.write(np.asarray(c.allele_number, dtype=np.uint32).tobytes())
        tab = np.array([(r.ref_start, r.length, r.strand, 0) for r in c.reads], dtype=np.int32)
        f.write(tab.tobytes())
        f.write(np.array([r.pool_off for r in c.reads], dtype=np.int64).tobytes())
        f.write(c.pool.tobytes())
