"""ctypes binding of oracle/pairhmm_oracle.c (banded pair-HMM forward probability).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by margin_amd.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liborc_pairhmm.so")


class Model(C.Structure):
    """StateMachine3 + NucleotideEmissions in log space (impl/stateMachine.c:507-519, inc/stateMachine.h)."""
    _fields_ = [(n, C.c_double) for n in ("match_continue", "match_from_gap_x", "match_from_gap_y", "gap_open_x", "gap_open_y",
                                          "gap_extend_x", "gap_extend_y", "gap_switch_to_x", "gap_switch_to_y")] + \
               [("e_match", C.c_double * 16), ("e_gap_x", C.c_double * 4), ("e_gap_y", C.c_double * 4)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pairhmm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s", "build/liborc_pairhmm.so"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
        L.pho_log_add.restype = dbl
        L.pho_log_add.argtypes = [dbl, dbl]
        L.pho_band.restype = C.c_int
        L.pho_band.argtypes = [vp, i64, i64, i64, i64, vp, vp]
        L.pho_forward_probability.restype = dbl
        L.pho_forward_probability.argtypes = [C.POINTER(Model), vp, i64, vp, i64, vp, i64, i64, C.c_int, C.c_int]
        L.pho_full_matrices.restype = C.c_int
        L.pho_full_matrices.argtypes = [C.POINTER(Model), vp, i64, vp, i64, i64, C.POINTER(dbl), C.POINTER(dbl), vp, vp]
        L.pho_test_cell.restype = None
        L.pho_test_cell.argtypes = [C.POINTER(Model), C.c_int, C.c_int, C.POINTER(dbl), C.POINTER(dbl)]
        L.pho_forward_batch.restype = None
        L.pho_forward_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, C.c_int, C.c_int, vp]
        L.pho_allele_read_supports.restype = None
        L.pho_allele_read_supports.argtypes = [C.POINTER(Model), C.POINTER(Model), i64, vp, vp, i64, vp, vp, vp, i64, i64, vp]
        L.pho_kmer_anchors.restype = i64
        L.pho_kmer_anchors.argtypes = [vp, i64, vp, i64, i64, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def log_add(x: float, y: float) -> float:
    return lib().pho_log_add(x, y)


def band(anchors, lx: int, ly: int, expansion: int):
    """(xmyL, xmyR) of every x+y diagonal (band_construct, impl/pairwiseAligner.c:175-226); anchors = [(x, y), ...]."""
    a = np.ascontiguousarray(np.asarray(anchors, dtype=np.int64).reshape(-1, 2))
    lo, hi = np.zeros(lx + ly + 1, np.int64), np.zeros(lx + ly + 1, np.int64)
    rc = lib().pho_band(_p(a), len(a), lx, ly, expansion, _p(lo), _p(hi))
    if rc != 0:
        raise ValueError(f"invalid band ({rc})")
    return lo, hi


def forward_probability(model: Model, sx: np.ndarray, sy: np.ndarray, anchors=(), expansion: int = 4, ragged_left=False, ragged_right=False) -> float:
    sx = np.ascontiguousarray(sx, dtype=np.uint8)
    sy = np.ascontiguousarray(sy, dtype=np.uint8)
    a = np.ascontiguousarray(np.asarray(anchors, dtype=np.int64).reshape(-1, 2))
    return lib().pho_forward_probability(C.byref(model), _p(sx), len(sx), _p(sy), len(sy), _p(a), len(a), expansion, int(ragged_left), int(ragged_right))


def full_matrices(model: Model, sx, sy, expansion: int = 2):
    sx = np.ascontiguousarray(sx, dtype=np.uint8)
    sy = np.ascontiguousarray(sy, dtype=np.uint8)
    tf, tb = C.c_double(), C.c_double()
    diag = np.zeros(len(sx) + len(sy) + 1)
    post = np.zeros((len(sx), len(sy)))
    rc = lib().pho_full_matrices(C.byref(model), _p(sx), len(sx), _p(sy), len(sy), expansion, C.byref(tf), C.byref(tb), _p(diag), _p(post))
    assert rc == 0
    return tf.value, tb.value, diag, post


def test_cell(model: Model, cx: int, cy: int):
    tf, tb = C.c_double(), C.c_double()
    lib().pho_test_cell(C.byref(model), cx, cy, C.byref(tf), C.byref(tb))
    return tf.value, tb.value


def forward_batch(models, pool, x_off, x_len, y_off, y_len, model_index=None, anchor_off=None, anchors=None, expansion=4, ragged_left=False,
                  ragged_right=False) -> np.ndarray:
    arr = (Model * len(models))(*models)
    n = len(x_off)
    out = np.zeros(n)
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    keep = [np.ascontiguousarray(x_off, dtype=np.int64), np.ascontiguousarray(x_len, dtype=np.int32), np.ascontiguousarray(y_off, dtype=np.int64),
            np.ascontiguousarray(y_len, dtype=np.int32), None if model_index is None else np.ascontiguousarray(model_index, dtype=np.uint8),
            None if anchor_off is None else np.ascontiguousarray(anchor_off, dtype=np.int64),
            None if anchors is None else np.ascontiguousarray(anchors, dtype=np.int64)]
    lib().pho_forward_batch(C.cast(arr, C.c_void_p), n, _p(pool), _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), _p(keep[4]), _p(keep[5]),
                            _p(keep[6]), expansion, int(ragged_left), int(ragged_right), _p(out))
    return out


def kmer_anchors(sx, sy, k: int = 20) -> np.ndarray:
    """getKmerAlignmentAnchors (impl/pairwiseAligner.c:1563-1627): int64 [n, 2] (x, y)"""
    sx = np.ascontiguousarray(sx, dtype=np.uint8)
    sy = np.ascontiguousarray(sy, dtype=np.uint8)
    out = np.zeros((max(len(sy), 1), 2), dtype=np.int64)
    n = lib().pho_kmer_anchors(_p(sx), len(sx), _p(sy), len(sy), k, _p(out))
    return out[:n].copy()


def allele_read_supports(forward_model: Model, reverse_model: Model, alleles, reads, read_forward_strand, expansion: int = 4,
                         sv_threshold: int = 512) -> np.ndarray:
    """alleles / reads: lists of uint8 symbol arrays of one bubble; returns float32 [n_alleles, n_reads] (bubbleGraph.c:1421-1464)."""
    al = [np.ascontiguousarray(a, dtype=np.uint8) for a in alleles]
    rd = [np.ascontiguousarray(r, dtype=np.uint8) for r in reads]
    ap = (C.c_void_p * len(al))(*[a.ctypes.data for a in al])
    rp = (C.c_void_p * len(rd))(*[r.ctypes.data for r in rd])
    alen = np.array([len(a) for a in al], dtype=np.int64)
    rlen = np.array([len(r) for r in rd], dtype=np.int64)
    st = np.ascontiguousarray(read_forward_strand, dtype=np.uint8)
    out = np.zeros((len(al), len(rd)), dtype=np.float32)
    lib().pho_allele_read_supports(C.byref(forward_model), C.byref(reverse_model), len(al), C.cast(ap, C.c_void_p), _p(alen), len(rd),
                                   C.cast(rp, C.c_void_p), _p(rlen), _p(st), expansion, sv_threshold, _p(out))
    return out
