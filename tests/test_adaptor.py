"""The reference-side adaptor (integration/stRPHmm_forwardBackward_adaptor.c): `void stRPHmm_forwardBackward(stRPHmm *)`
implemented by flatten -> mrp_fb_run -> scatter back.  margin itself cannot be built here (sonLib / htslib submodules
are empty), so the very same source file is compiled against the oracle's linked-list hmm -- whose structs mirror
inc/margin.h field for field -- through tests/adaptor_binding/margin_as_oracle.h, and every post-condition of the
replaced function is checked bit for bit against the oracle's own stRPHmm_forwardBackward."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from margin_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("cell_forward", "cell_backward", "merge_forward", "merge_backward", "col_total")


def build_adaptor(orc):
    out = os.path.join(ROOT, "oracle", "build", "libadaptor_test.so")
    src = [os.path.join(ROOT, "integration", "stRPHmm_forwardBackward_adaptor.c"), os.path.join(ROOT, "tests", "adaptor_binding", "driver.c")]
    liborc = orc.build()
    libmrp = capi.LIB_PATH
    newest = max(os.path.getmtime(p) for p in src + [os.path.join(ROOT, "tests", "adaptor_binding", "margin_as_oracle.h"), liborc, libmrp])
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror", "-shared",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "adaptor_binding"),
                               '-DMRP_ADAPTOR_BINDING_HEADER="margin_as_oracle.h"', "-o", out] + src +
                              [liborc, libmrp, "-Wl,-rpath," + os.path.dirname(liborc), "-Wl,-rpath," + os.path.dirname(libmrp), "-lm"])
    L = C.CDLL(out)
    P = C.POINTER
    L.adp_test_poison.argtypes = [P(orc.Hmm)]
    L.adp_test_forward_backward.argtypes = [P(orc.Hmm)]
    L.adp_test_forward_backward_many.argtypes = [P(P(orc.Hmm)), C.c_int64]
    L.adp_test_register.argtypes = [C.c_void_p, P(C.c_void_p), C.c_int64]
    L.adp_test_unregister.argtypes = [C.c_void_p]
    L.adp_test_set_min_cells.argtypes = [C.c_int64]
    for f in ("adp_test_poison", "adp_test_forward_backward", "adp_test_forward_backward_many", "adp_test_register", "adp_test_unregister", "adp_test_cleanup",
              "adp_test_set_min_cells"):
        getattr(L, f).restype = None
    return L


def test_adaptor_compiles_against_the_binding_and_exports_the_seam(orc):
    """CPU: the adaptor builds warning-free against the oracle binding and defines the function it replaces."""
    L = build_adaptor(orc)
    for name in ("stRPHmm_forwardBackward", "stRPHmm_forwardBackwardMany", "mrpAdaptor_registerProfileSeqs", "mrpAdaptor_unregister",
                 "mrpAdaptor_threadCleanup", "mrpAdaptor_setMinCells"):
        assert hasattr(L, name)


def _same(exp, got):
    for k in FIELDS:
        x, y = np.asarray(exp[k]), np.asarray(got[k])
        assert x.shape == y.shape and ((x == y) | (np.isneginf(x) & np.isneginf(y))).all(), k
    assert exp["hmm_forward"] == got["hmm_forward"] and exp["hmm_backward"] == got["hmm_backward"]


@pytest.mark.gpu
@pytest.mark.parametrize("registered,min_cells", [(True, 0), (False, 0), (True, 4096)])
def test_adaptor_reproduces_every_post_condition(orc, registered, min_cells):
    """Inside the oracle's phasing driver, right after each of ITS stRPHmm_forwardBackward calls (coordination.c:312 at every
    merge level, bubbleGraph.c:2749 with the ancestor model): the results are saved, every field the function has to set is
    overwritten with NaN, the adaptor sweeps the live linked-list hmm on the GPU, and the graph must hold the oracle's values
    again -- cells in list order, merge cells (unreachable ones -inf), column totals, the hmm's totals.  The driver then
    carries on (prune, next level) with the adaptor's values: the final haplotypes equal an undisturbed run's."""
    L = orc.lib()
    A = build_adaptor(orc)
    # min_cells = 0: every hmm on the device; 4096 (the adaptor's default): the small ones stay on the CPU body it replaces
    A.adp_test_set_min_cells(min_cells)
    chunk = synth.make_ont_chunk(seed=41, region_bp=70_000, n_sites=140, coverage=26, allele_choices=(2, 3), allele_probs=(0.8, 0.2))
    rng = np.random.default_rng(5)
    chunk.sub = rng.integers(0, 200, size=chunk.sub.shape).astype(np.uint16)
    chunk.prior = rng.integers(0, 60, size=chunk.prior.shape).astype(np.uint16)
    pd = synth.shipped_phase_params()
    oc = orc.OracleChunk(chunk)
    undisturbed = oc.phase(pd)
    if registered:
        A.adp_test_register(oc.ref, oc.seq_array(), len(oc.seqs))
    seen = []

    def obs(hmm_ptr, _user):
        exp = orc.flatten(hmm_ptr, oc.pool_off)
        A.adp_test_poison(hmm_ptr)
        assert np.isnan(orc.flatten(hmm_ptr, oc.pool_off)["cell_forward"]).all()
        A.adp_test_forward_backward(hmm_ptr)
        got = orc.flatten(hmm_ptr, oc.pool_off)
        _same(exp, got)
        seen.append((int(exp["n_columns"]), int(np.isneginf(exp["merge_forward"]).sum())))

    cb = orc.FB_OBSERVER(obs)
    L.orc_set_fb_observer(cb, None)
    try:
        params = orc.make_params(pd)
        final = C.POINTER(orc.Hmm)()
        gf = L.orc_phase_profile_seqs(oc.seq_array(), oc.strands.ctypes.data, len(oc.seqs), C.byref(params), None)
    finally:
        L.orc_set_fb_observer(C.cast(None, orc.FB_OBSERVER), None)
    orc.check_error()
    g = gf.contents
    n = int(g.length)
    assert [g.haplotypeString1[i] for i in range(n)] == list(undisturbed["hap1"])
    assert [int(g.reads1[i]) for i in range(g.nReads1)] == undisturbed["reads1"]
    L.orc_genome_fragment_destroy(gf)
    assert len(seen) == undisturbed["fb_calls"] and max(k for k, _ in seen) > 50
    if registered:
        A.adp_test_unregister(oc.ref)
    A.adp_test_cleanup()
    oc.close()


@pytest.mark.gpu
def test_adaptor_batched_variant(orc):
    """stRPHmm_forwardBackwardMany: the pruned hmms of getRPHmms of two strands swept in ONE device batch equal the oracle's
    sweeps of the same hmms, one by one."""
    L = orc.lib()
    A = build_adaptor(orc)
    A.adp_test_set_min_cells(0)
    chunk = synth.make_ont_chunk(seed=43, region_bp=60_000, n_sites=120, coverage=24)
    pd = dict(synth.shipped_phase_params(), includeAncestorSubProb=0)
    oc = orc.OracleChunk(chunk)
    A.adp_test_register(oc.ref, oc.seq_array(), len(oc.seqs))
    hmms = []
    for strand in (1, 0):
        hmms += oc.get_rp_hmms(orc.make_params(pd), [i for i, r in enumerate(chunk.reads) if r.strand == strand])
    assert len(hmms) >= 2
    exp = []
    for h in hmms:
        L.orc_hmm_forwardBackward(h)
        exp.append(orc.flatten(h, oc.pool_off))
        A.adp_test_poison(h)
    arr = (C.POINTER(orc.Hmm) * len(hmms))(*hmms)
    A.adp_test_forward_backward_many(arr, len(hmms))
    for h, e in zip(hmms, exp):
        _same(e, orc.flatten(h, oc.pool_off))
    A.adp_test_unregister(oc.ref)
    A.adp_test_cleanup()
    oc.close()
