"""ctypes binding of the CPU oracle (oracle/rphmm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package (margin_amd).  The flat job dictionaries produced here use exactly the field
names of ``mrp_hmm_job`` in include/margin_rphmm.h so a test can hand the same arrays to the HIP
path and compare.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liborc_rphmm.so")


def build(force: bool = False) -> str:
    """Compile the oracle with the reference's flags (CMakeLists.txt:5: -O3, plus -mpopcnt -fopenmp)."""
    src = os.path.join(_HERE, "rphmm_oracle.c")
    hdr = os.path.join(_HERE, "rphmm_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [("maxNotSumTransitions", C.c_int32),
                ("minPartitionsInAColumn", C.c_int64),
                ("maxPartitionsInAColumn", C.c_int64),
                ("minPosteriorProbabilityForPartition", C.c_double),
                ("maxCoverageDepth", C.c_int64),
                ("minReadCoverageToSupportPhasingBetweenHeterozygousSites", C.c_int64),
                ("includeInvertedPartitions", C.c_int32),
                ("roundsOfIterativeRefinement", C.c_int64),
                ("includeAncestorSubProb", C.c_int32)]


class Hmm(C.Structure):
    _fields_ = [("ref", C.c_void_p), ("refStart", C.c_int64), ("refLength", C.c_int64),
                ("profileSeqs", C.c_void_p), ("nProfileSeqs", C.c_int64), ("columnNumber", C.c_int64),
                ("maxDepth", C.c_int64), ("firstColumn", C.c_void_p), ("lastColumn", C.c_void_p),
                ("parameters", C.c_void_p), ("forwardLogProb", C.c_double), ("backwardLogProb", C.c_double)]


class GenomeFragment(C.Structure):
    _fields_ = [("reference", C.c_void_p), ("refStart", C.c_uint64), ("length", C.c_uint64),
                ("reads1", C.POINTER(C.c_int64)), ("reads2", C.POINTER(C.c_int64)),
                ("nReads1", C.c_int64), ("nReads2", C.c_int64),
                ("genotypeString", C.POINTER(C.c_uint64)), ("haplotypeString1", C.POINTER(C.c_uint64)),
                ("haplotypeString2", C.POINTER(C.c_uint64)), ("ancestorString", C.POINTER(C.c_uint64)),
                ("readsSupportingHaplotype1", C.POINTER(C.c_uint64)),
                ("readsSupportingHaplotype2", C.POINTER(C.c_uint64)),
                ("genotypeProbs", C.POINTER(C.c_float)), ("haplotypeProbs1", C.POINTER(C.c_float)),
                ("haplotypeProbs2", C.POINTER(C.c_float))]


FB_OBSERVER = C.CFUNCTYPE(None, C.POINTER(Hmm), C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp, i64, u64, dbl, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int32
    P = C.POINTER
    sig = {
        "orc_last_error": (C.c_char_p, []),
        "orc_clear_error": (None, []),
        "orc_makeAcceptMask": (u64, [u64]),
        "orc_mergePartitionsOrMasks": (u64, [u64, u64, u64, u64]),
        "orc_maskPartition": (u64, [u64, u64]),
        "orc_invertPartition": (u64, [u64, u64]),
        "orc_seqInHap1": (C.c_int, [u64, i64]),
        "orc_flipAReadsPartition": (u64, [u64, u64]),
        "orc_popcount64": (C.c_int, [u64]),
        "orc_logAddExact": (dbl, [dbl, dbl]),
        "orc_logAddP": (dbl, [dbl, dbl, C.c_int]),
        "orc_reference_create": (vp, [C.c_char_p, i64, vp, vp, vp]),
        "orc_reference_destroy": (None, [vp]),
        "orc_profile_seq_create": (vp, [vp, C.c_char_p, i64, i64, i64, vp]),
        "orc_profile_seq_destroy": (None, [vp]),
        "orc_calculateCountBitVectors": (P(u64), [vp, vp, u64, u64, u64]),
        "orc_getLogProbOfAllele": (u64, [vp, u64, u64, u64, u64]),
        "orc_emission_raw": (dbl, [vp, vp, i64, i64, i64, u64, C.c_int]),
        "orc_hmm_construct": (P(Hmm), [vp, P(Params)]),
        "orc_hmm_destruct": (None, [P(Hmm), C.c_int]),
        "orc_hmm_overlapOnReference": (C.c_int, [P(Hmm), P(Hmm)]),
        "orc_hmm_fuse": (P(Hmm), [P(Hmm), P(Hmm)]),
        "orc_hmm_alignColumns": (None, [P(Hmm), P(Hmm)]),
        "orc_hmm_createCrossProductOfTwoAlignedHmm": (P(Hmm), [P(Hmm), P(Hmm)]),
        "orc_hmm_forwardBackward": (None, [P(Hmm)]),
        "orc_hmm_prune": (None, [P(Hmm)]),
        "orc_hmm_forwardTraceBack": (P(vp), [P(Hmm), P(i64)]),
        "orc_hmm_splitWherePhasingIsUncertain": (P(P(Hmm)), [P(Hmm), P(i64)]),
        "orc_hmm_split": (P(Hmm), [P(Hmm), i64]),
        "orc_getRPHmms": (P(P(Hmm)), [P(vp), i64, P(Params), P(i64)]),
        "orc_filterReadsByCoverageDepth": (None, [P(vp), i64, P(Params), P(vp), P(i64), P(vp), P(i64)]),
        "orc_tilingPathCount": (i64, [P(vp), i64, P(Params)]),
        "orc_genome_fragment_construct": (P(GenomeFragment), [P(Hmm), P(vp), i64]),
        "orc_genome_fragment_destroy": (None, [P(GenomeFragment)]),
        "orc_phase_profile_seqs": (P(GenomeFragment), [P(vp), vp, i64, P(Params), P(P(Hmm))]),
        "orc_set_fb_observer": (None, [FB_OBSERVER, vp]),
        "orc_fb_timer_reset": (None, []),
        "orc_fb_timer_seconds": (dbl, []),
        "orc_fb_timer_calls": (i64, []),
        "orc_hmm_flat_sizes": (None, [P(Hmm), P(i64)]),
        "orc_hmm_flatten": (None, [P(Hmm)] + [vp] * 20),
        "free": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        if name == "free":
            continue
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check_error():
    msg = lib().orc_last_error()
    if msg:
        lib().orc_clear_error()
        raise RuntimeError("oracle: " + msg.decode())


def make_params(d: dict) -> Params:
    p = Params()
    for k, v in d.items():
        setattr(p, k, v)
    return p


def flatten(hmm_ptr, read_pool_off: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """Flatten an oracle hmm (with whatever forward/backward values it currently holds)."""
    L = lib()
    sizes = (C.c_int64 * 4)()
    L.orc_hmm_flat_sizes(hmm_ptr, sizes)
    K, nC, nM, nD = (int(x) for x in sizes)
    a = dict(
        col_ref_start=np.zeros(K, np.int32), col_length=np.zeros(K, np.int32), col_depth=np.zeros(K, np.int32),
        col_cell_off=np.zeros(K + 1, np.int64), col_read_off=np.zeros(K + 1, np.int64),
        read_byte_off=np.zeros(max(nD, 1), np.int64), read_ids=np.zeros(max(nD, 1), np.int64),
        partition=np.zeros(nC, np.uint64), mask_from=np.zeros(max(K - 1, 1), np.uint64),
        mask_to=np.zeros(max(K - 1, 1), np.uint64), mcol_cell_off=np.zeros(K, np.int64),
        merge_from=np.zeros(max(nM, 1), np.uint64), merge_to=np.zeros(max(nM, 1), np.uint64),
        cell_next=np.zeros(nC, np.uint32), cell_prev=np.zeros(nC, np.uint32),
        cell_forward=np.zeros(nC, np.float64), cell_backward=np.zeros(nC, np.float64),
        merge_forward=np.zeros(max(nM, 1), np.float64), merge_backward=np.zeros(max(nM, 1), np.float64),
        col_total=np.zeros(K, np.float64))
    order = ["col_ref_start", "col_length", "col_depth", "col_cell_off", "col_read_off", "read_byte_off",
             "read_ids", "partition", "mask_from", "mask_to", "mcol_cell_off", "merge_from", "merge_to",
             "cell_next", "cell_prev", "cell_forward", "cell_backward", "merge_forward", "merge_backward",
             "col_total"]
    L.orc_hmm_flatten(hmm_ptr, *[a[k].ctypes.data_as(C.c_void_p) for k in order])
    a["read_byte_off"] = a["read_byte_off"][:nD]
    a["read_ids"] = a["read_ids"][:nD]
    a["mask_from"] = a["mask_from"][:K - 1]
    a["mask_to"] = a["mask_to"][:K - 1]
    a["merge_from"] = a["merge_from"][:nM]
    a["merge_to"] = a["merge_to"][:nM]
    a["merge_forward"] = a["merge_forward"][:nM]
    a["merge_backward"] = a["merge_backward"][:nM]
    if read_pool_off is not None:
        a["read_byte_off"] = a["read_byte_off"] + read_pool_off[a["read_ids"]]
    h = hmm_ptr.contents
    a["n_columns"] = K
    a["hmm_forward"] = float(h.forwardLogProb)
    a["hmm_backward"] = float(h.backwardLogProb)
    a["ref_start"] = int(h.refStart)
    a["ref_length"] = int(h.refLength)
    return a


class OracleChunk:
    """The oracle's view of a synthetic chunk: an stReference and one stProfileSeq per read."""

    def __init__(self, chunk):
        L = lib()
        self.chunk = chunk
        an = np.ascontiguousarray(chunk.allele_number, dtype=np.uint32)
        sub = np.ascontiguousarray(chunk.sub, dtype=np.uint16)
        prior = np.ascontiguousarray(chunk.prior, dtype=np.uint16)
        self.ref = L.orc_reference_create(b"ref", chunk.n_sites, an.ctypes.data, sub.ctypes.data, prior.ctypes.data)
        self.seqs = []
        pool = np.ascontiguousarray(chunk.pool)
        self._pool = pool
        for i, r in enumerate(chunk.reads):
            s = L.orc_profile_seq_create(self.ref, r.name.encode(), i, r.ref_start, r.length,
                                         pool.ctypes.data + r.pool_off)
            self.seqs.append(s)
        self.pool_off = np.array([r.pool_off for r in chunk.reads], dtype=np.int64)
        self.strands = np.array([r.strand for r in chunk.reads], dtype=np.uint8)

    def seq_array(self, idx=None):
        idx = range(len(self.seqs)) if idx is None else idx
        arr = (C.c_void_p * max(len(list(idx)), 1))()
        for j, i in enumerate(idx):
            arr[j] = self.seqs[i]
        return arr

    def close(self):
        L = lib()
        for s in self.seqs:
            L.orc_profile_seq_destroy(s)
        self.seqs = []
        if self.ref:
            L.orc_reference_destroy(self.ref)
            self.ref = None

    # ---- drivers ------------------------------------------------------------------------
    def get_rp_hmms(self, params: Params, idx=None):
        """coordination.c:490 getRPHmms over the given reads; returns list of hmm pointers."""
        L = lib()
        idx = list(range(len(self.seqs))) if idx is None else list(idx)
        arr = self.seq_array(idx)
        n_out = C.c_int64(0)
        out = L.orc_getRPHmms(arr, len(idx), C.byref(params), C.byref(n_out))
        check_error()
        # the hmms keep a POINTER to the parameters (stRPHmm.parameters): the ctypes object has to outlive them, also when the
        # caller passed a temporary
        self._params_alive = getattr(self, "_params_alive", [])
        self._params_alive.append(params)
        return [out[i] for i in range(n_out.value)]

    def phase(self, params_dict: dict, capture_jobs: bool = False, keep_final: bool = False,
              on_job: Optional[Callable[[Dict[str, np.ndarray]], None]] = None):
        """bubbleGraph.c:2673 phasing driver.  Returns dict with haplotype strings, read sets,
        optionally every forward/backward job (flattened, with the oracle's results)."""
        L = lib()
        params = make_params(params_dict)
        jobs: List[Dict[str, np.ndarray]] = []

        def _obs(hmm_ptr, _user):
            j = flatten(hmm_ptr, self.pool_off)
            pr = C.cast(hmm_ptr.contents.parameters, C.POINTER(Params)).contents
            j["flags"] = (1 if pr.maxNotSumTransitions else 0) | (2 if pr.includeAncestorSubProb else 0)
            if on_job is not None:
                on_job(j)
            else:
                jobs.append(j)

        cb = FB_OBSERVER(_obs)
        if capture_jobs or on_job is not None:
            L.orc_set_fb_observer(cb, None)
        final = C.POINTER(Hmm)()
        L.orc_fb_timer_reset()
        try:
            gf = L.orc_phase_profile_seqs(self.seq_array(), self.strands.ctypes.data, len(self.seqs),
                                          C.byref(params), C.byref(final) if keep_final else None)
        finally:
            L.orc_set_fb_observer(C.cast(None, FB_OBSERVER), None)
        check_error()
        g = gf.contents
        n = int(g.length)
        res = dict(
            ref_start=int(g.refStart), length=n,
            reads1=[int(g.reads1[i]) for i in range(g.nReads1)],
            reads2=[int(g.reads2[i]) for i in range(g.nReads2)],
            hap1=np.array([g.haplotypeString1[i] for i in range(n)], dtype=np.uint64),
            hap2=np.array([g.haplotypeString2[i] for i in range(n)], dtype=np.uint64),
            genotype=np.array([g.genotypeString[i] for i in range(n)], dtype=np.uint64),
            ancestor=np.array([g.ancestorString[i] for i in range(n)], dtype=np.uint64),
            genotype_probs=np.array([g.genotypeProbs[i] for i in range(n)], dtype=np.float32),
            hap_probs1=np.array([g.haplotypeProbs1[i] for i in range(n)], dtype=np.float32),
            hap_probs2=np.array([g.haplotypeProbs2[i] for i in range(n)], dtype=np.float32),
            support1=np.array([g.readsSupportingHaplotype1[i] for i in range(n)], dtype=np.uint64),
            support2=np.array([g.readsSupportingHaplotype2[i] for i in range(n)], dtype=np.uint64),
            fb_seconds=float(L.orc_fb_timer_seconds()), fb_calls=int(L.orc_fb_timer_calls()), jobs=jobs)
        L.orc_genome_fragment_destroy(gf)
        if keep_final:
            res["final_hmm"] = final
        return res
