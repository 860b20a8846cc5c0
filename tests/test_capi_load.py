"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from margin_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "margin_rphmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrp_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert set(capi.EXPORTED_SYMBOLS) <= set(declared)


def test_no_torch_types_and_c_linkage():
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    for s in _declared_symbols():
        assert s in exported            # unmangled: extern "C"
    assert not [s for s in exported if "torch" in s.lower() or "at::" in s]


def test_fails_loudly_without_a_device():
    lib = capi.load()
    if lib.mrp_device_count() > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    rc = lib.mrp_context_create(0, C.byref(h))
    assert rc == capi.MRP_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.mrp_last_error()
    with pytest.raises(capi.MrpError):
        capi.Context(0)


def test_product_package_does_not_touch_the_oracle():
    """The shipped path may not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "margin_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"liborc", r"\borc_[a-z]", r"rphmm_oracle", r"oracle/"):
                    assert not re.search(pat, text, flags=re.M), (pat, os.path.join(dirpath, f))
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "liborc" not in out


def test_host_worker_pool_runs_every_index_once():
    """mrp_pool_run (the persistent pool behind the host-side parallel loops): concurrent callers, grains, nesting."""
    import threading
    lib = capi.load()
    CB = C.CFUNCTYPE(None, C.c_int64, C.c_void_p)
    lib.mrp_pool_run.argtypes = [C.c_int64, C.c_int64, CB, C.c_void_p]
    lib.mrp_pool_run.restype = None
    errors = []

    def caller(seed):
        for r in range(30):
            n = 1 + (seed * 31 + r * 17) % 200
            hits = [0] * n
            lock = threading.Lock()

            def body(i, _arg):
                with lock:
                    hits[i] += 1
            cb = CB(body)
            lib.mrp_pool_run(n, 1 + r % 4, cb, None)
            if hits != [1] * n:
                errors.append((seed, r, n))

    def big(seed):  # loops longer than the pool's ranges: every range, whole grains, the clamped last one
        for n, grain in ((1, 7), (15, 1), (16, 1), (17, 3), (255, 16), (1000, 64), (4097, 1), (5000, 7)):
            hits = [0] * n
            lock = threading.Lock()

            def body(i, _arg):
                with lock:
                    hits[i] += 1
            cb = CB(body)
            lib.mrp_pool_run(n, grain, cb, None)
            if hits != [1] * n:
                errors.append((seed, "big", n, grain))

    threads = [threading.Thread(target=caller, args=(s,)) for s in range(3)] + [threading.Thread(target=big, args=(9,))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


def test_batch_shares_of_a_call(monkeypatch):
    """mrp_phase_group_assign (rphmm_host.c): which concurrent batch every chunk of an mrp_phase_reads_many call goes to -- and, the
    same function, in which group a work queue uploads it.  Equal shares for small calls and small chunks, graded ones
    (2 : 3 : 4 : 5 : 5 ...) for many large chunks, MRP_GROUP_WEIGHTS on request; every chunk gets a batch below G."""
    lib = capi.load()
    lib.mrp_phase_group_assign.argtypes = [C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_uint8)]
    lib.mrp_phase_group_assign.restype = None
    monkeypatch.delenv("MRP_GROUP_WEIGHTS", raising=False)

    def shares(n, g, sites):
        buf = (C.c_uint8 * (n + 1))()
        lib.mrp_phase_group_assign(n, g, sites, buf)
        got = list(buf)[:n]
        assert all(0 <= x < g for x in got)
        return [got.count(q) for q in range(g)], got

    cnt, got = shares(1152, 8, 1152 * 2000)          # the headline call: graded
    assert cnt[0] < cnt[1] < cnt[2] < cnt[3] and max(cnt[3:]) - min(cnt[3:]) <= 1 and cnt[0] * 2 < cnt[-1] and sum(cnt) == 1152
    assert abs(cnt[0] / 1152 - 2 / 34) < 0.01 and abs(cnt[-1] / 1152 - 5 / 34) < 0.01
    assert got[:8] == list(range(8))                 # (every batch has a chunk among the first G: the pattern goes round by round)
    cnt, _ = shares(640, 8, 640 * 130)               # chunks of a few hundred sites: equal
    assert max(cnt) - min(cnt) <= 1
    cnt, _ = shares(64, 8, 64 * 2000)                # fewer than 16 chunks per batch: equal
    assert max(cnt) - min(cnt) <= 1
    cnt, _ = shares(300, 2, 300 * 2000)              # fewer than four batches: equal
    assert max(cnt) - min(cnt) <= 1
    monkeypatch.setenv("MRP_GROUP_WEIGHTS", "1:3")
    cnt, _ = shares(400, 2, 400 * 130)
    assert cnt == [100, 300]
