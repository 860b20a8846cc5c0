"""Shared helpers for the parity tests (test infrastructure; may use the oracle)."""
import numpy as np

from margin_amd import capi


def job_flags(flat):
    return int(flat["flags"])


def run_jobs_on_gpu(ctx, dchunk, flats, use_indices=True, flags_override=None):
    jobs = [capi.Job(dchunk, f, job_flags(f) if flags_override is None else flags_override, use_indices) for f in flats]
    capi.fb_run(ctx, jobs)
    return [j.results() for j in jobs]


def assert_job_equal(flat, res, exact=True, atol=0.0):
    names = [("cell_forward", "cell_forward"), ("cell_backward", "cell_backward"), ("merge_forward", "merge_forward"),
             ("merge_backward", "merge_backward"), ("col_total", "col_total")]
    for a, b in names:
        x, y = np.asarray(flat[a], dtype=np.float64), np.asarray(res[b], dtype=np.float64)
        assert x.shape == y.shape, (a, x.shape, y.shape)
        if exact:
            same = (x == y) | (np.isneginf(x) & np.isneginf(y))
            assert same.all(), (a, int((~same).sum()), x[~same][:5], y[~same][:5])
        else:
            fin = np.isfinite(x)
            assert (np.isneginf(x) == np.isneginf(y)).all(), a
            assert np.allclose(x[fin], y[fin], rtol=0, atol=atol), (a, np.abs(x[fin] - y[fin]).max())
    for a in ("hmm_forward", "hmm_backward"):
        x, y = float(np.asarray(flat[a]).reshape(-1)[0]), float(np.asarray(res[a]).reshape(-1)[0])
        if exact:
            assert x == y, (a, x, y)
        else:
            assert abs(x - y) <= atol, (a, x, y)


def posteriors(flat_like, f, b, total):
    """column.c:177-193 for every cell."""
    K = int(flat_like["n_columns"])
    off = flat_like["col_cell_off"]
    out = np.empty_like(f)
    for k in range(K):
        s = slice(int(off[k]), int(off[k + 1]))
        out[s] = np.minimum(1.0, np.exp(f[s] + b[s] - total[k]))
    return out
