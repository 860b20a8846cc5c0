"""The host-side work queue over the GPUs of a node (mrp_queue.cpp; reference: phase.c:257-263 chunk order by estimated
depth, largest first, and phase.c:276-279 "schedule(dynamic,1)").  CPU: the order and the queue itself with stand-in
workers.  GPU: two workers sharing device 0 phase a set of chunks; results equal the oracle's whatever worker took them."""
import numpy as np
import pytest

from margin_amd import capi, synth


def test_queue_order_largest_first_ties_in_input_order():
    cost = np.array([5, 9, 9, 1, 7, 9, 0, 3], dtype=np.int64)
    order, batch = capi.queue_plan(cost, 3)
    assert order.tolist() == [1, 2, 5, 4, 0, 7, 3, 6]           # phase.c:257-263, stable on ties
    assert batch[order].tolist() == [0, 0, 0, 1, 1, 1, 2, 2]   # consecutive chunks of that order travel together
    order1, batch1 = capi.queue_plan(cost, 1)
    assert order1.tolist() == order.tolist() and sorted(batch1.tolist()) == list(range(8))
    empty, _ = capi.queue_plan(np.zeros(0, dtype=np.int64), 4)
    assert len(empty) == 0


@pytest.mark.parametrize("n_devices,lanes,per_batch", [(1, 1, 1), (2, 1, 1), (3, 1, 2), (2, 4, 1), (4, 4, 4)])
def test_queue_dry_run_is_complete_ordered_and_dynamic(n_devices, lanes, per_batch):
    """Every chunk is taken exactly once; a worker takes its batches in the queue's order (largest first); the schedule is
    dynamic: with one very expensive batch, the worker that holds it takes at most the batch it had already reserved
    (a lane takes its next batch when it starts on the current one, as the real worker does to upload it meanwhile) while the
    rest drain the queue."""
    n_workers = n_devices * lanes
    rng = np.random.default_rng(n_workers * 10 + per_batch)
    cost = rng.integers(1, 50, size=97).astype(np.int64)
    worker, seq = capi.queue_dry_run(n_devices, lanes, cost, per_batch, usec_per_cost=2.0)
    assert (worker >= 0).all() and (worker < n_workers).all()
    assert sorted(seq.tolist()) == list(range(len(cost)))
    order, batch = capi.queue_plan(cost, per_batch)
    for w in range(n_workers):
        mine = [i for i in order if worker[i] == w]          # in queue order
        assert [seq[i] for i in mine] == sorted(seq[i] for i in mine)
    # chunks of one batch go to one worker
    for b in range(int(batch.max()) + 1):
        assert len(set(worker[batch == b].tolist())) == 1
    if n_workers >= 2:
        big = cost.copy()
        big[13] = 40_000  # 80 ms at 2 us per unit: the others are through long before
        worker, _ = capi.queue_dry_run(n_devices, lanes, big, 1, usec_per_cost=2.0)
        holder = worker[13]
        assert (worker == holder).sum() <= 2  # the expensive chunk heads the queue; its worker holds only the one batch it took ahead


def test_whole_genome_queue_on_eight_devices_is_balanced():
    """BASELINE.json configs[3]: ~31 000 chunks of ~100 kb over the 8 devices of a node, four lanes each = 32
    pulling threads.  With the library's own batch sizes (chunks_per_batch = 0; cut by units, here one striped batch per pulling thread) every chunk is taken once and no device carries more than 2 % above the mean of the estimated cost
    (het-sites x reads) -- the stand-in workers sleep in proportion to it, so the hand-out is the dynamic one of a real run."""
    rng = np.random.default_rng(31_000)
    n = 31_000
    sites = np.clip(rng.normal(130, 25, size=n), 20, 400)          # het sites per chunk
    depth = np.clip(rng.normal(30, 6, size=n), 5, 64)              # coverage varies along the genome
    cost = (sites * depth * 2.0).astype(np.int64)                  # reads x het sites each spans (~2 x depth x sites / ...): units
    lanes, devices = 4, 8
    usec = 2.0e5 * devices * lanes / float(cost.sum())             # about 0.2 s of stand-in work per pulling thread
    worker, seq = capi.queue_dry_run(devices, lanes, cost, 0, usec_per_cost=usec)
    assert sorted(seq.tolist()) == list(range(n))
    per_device = np.bincount(worker // lanes, weights=cost.astype(np.float64), minlength=devices)
    assert (per_device > 0).all()
    imbalance = per_device.max() / per_device.mean() - 1.0
    assert imbalance <= 0.02, (imbalance, per_device.tolist())
    # this queue (2.4e8 units) is below 1 280 yardstick chunks per device: ONE striped batch per device, one lane each
    assert len(set(worker.tolist())) == devices and (worker % lanes == 0).all()
    # one device: the batches follow the queue's order and are cut by UNITS (192 chunks of 60 000 units = 1.152e7 per batch: ~1 480 of
    # these small chunks), not by chunk count
    order, batch = capi.queue_plan(cost, 0)
    per_batch = np.bincount(batch, weights=cost.astype(np.float64))
    assert (np.diff(batch[order]) >= 0).all() and (np.diff(cost[order]) <= 0).all()
    n_batches = int(np.ceil(cost.sum() / 1.152e7))
    assert len(per_batch) == n_batches
    assert (np.abs(per_batch[:-1] / (cost.sum() / n_batches) - 1.0) < 0.01).all() and 0.5 * 1.152e7 <= per_batch[-1] <= 1.5 * 1.152e7
    assert np.bincount(batch).min() > 800  # (the first batches hold the most expensive chunks: fewer of them)


def test_long_queue_on_eight_devices_takes_the_guided_schedule_and_stays_balanced():
    """More than 1 280 yardstick chunks (60 000 units) per device: the multi-device branch of the plan -- batches of at most 192
    yardstick chunks, shrinking towards the end (what is left / twice the lanes, at least a quarter batch: a lane holds the batch
    it phases and the one it took ahead) -- handed out dynamically to 8 devices x 4 lanes.  The stand-in calls sleep in
    proportion to their units; every lane of every device takes work and the devices end within 4 % of the mean cost."""
    rng = np.random.default_rng(8)
    n = 90_000
    sites = np.clip(rng.normal(130, 25, size=n), 20, 400)
    depth = np.clip(rng.normal(30, 6, size=n), 5, 64)
    cost = (sites * depth * 2.0).astype(np.int64)
    lanes, devices = 4, 8
    assert cost.sum() > devices * 1280 * 60_000
    usec = 4.0e5 * devices * lanes / float(cost.sum())  # about 0.4 s of stand-in work per lane
    worker, seq = capi.queue_dry_run(devices, lanes, cost, 0, usec_per_cost=usec)
    assert sorted(seq.tolist()) == list(range(n)) and (worker >= 0).all()
    assert len(set(worker.tolist())) == devices * lanes
    per_device = np.bincount(worker // lanes, weights=cost.astype(np.float64), minlength=devices)
    imbalance = per_device.max() / per_device.mean() - 1.0
    assert imbalance <= 0.04, (imbalance, per_device.tolist())
    # the first batches are fixed: batch b of the plan (the most expensive chunks) starts on device b % 8, lane b // 8
    order = np.argsort(-cost, kind="stable")
    assert [int(worker[order[0]]), int(worker[order[-1]] >= 0)] == [0, 1]
    first_of_worker = {int(w): int(np.flatnonzero(worker[order] == w)[0]) for w in set(worker.tolist())}
    starts = sorted(first_of_worker.items(), key=lambda kv: kv[1])
    assert [w for w, _ in starts[:devices]] == [d * lanes for d in range(devices)]
    # no batch above 192 yardstick chunks; the tail is made of quarter batches, not dust: a worker's share of a run of the cost order
    full = 192 * 60_000
    wo = worker[order]
    cuts = np.flatnonzero(np.diff(wo) != 0) + 1
    sizes = np.array([c.sum() for c in np.split(cost[order].astype(np.float64), cuts)])
    assert sizes.max() <= 2.04 * full   # (two consecutive batches of one lane show as one run)
    assert np.sort(sizes)[1] >= 0.2 * full and np.median(sizes[-32:]) <= 0.3 * full


def test_queue_rejects_bad_arguments():
    L = capi.load()
    cost = np.ones(4, dtype=np.int64)
    w = np.zeros(4, dtype=np.int32)
    assert L.mrp_queue_dry_run(0, 1, 4, cost.ctypes.data, 1, 0.0, w.ctypes.data, None) == capi.MRP_ERR_ARG
    assert L.mrp_queue_dry_run(capi.MAX_QUEUE_DEVICES + 1, 1, 4, cost.ctypes.data, 1, 0.0, w.ctypes.data, None) == capi.MRP_ERR_ARG
    assert L.mrp_queue_dry_run(1, 5, 4, cost.ctypes.data, 1, 0.0, w.ctypes.data, None) == capi.MRP_ERR_ARG
    # without a device the real queue fails loudly: no CPU fallback
    if L.mrp_device_count() == 0:
        import ctypes as C
        q = C.c_void_p()
        dev = (C.c_int32 * 1)(0)
        assert L.mrp_queue_create(C.cast(dev, C.c_void_p), 1, C.byref(q)) == capi.MRP_ERR_NO_DEVICE


@pytest.mark.gpu
def test_two_workers_on_one_device_phase_every_chunk(orc):
    """mrp_queue_phase_chunks with the device listed twice: two host threads, two contexts, batches of 3 chunks pulled
    from the shared queue; every chunk's result sits at its own position and equals the oracle's; a second call on the
    same queue object reuses the workers' contexts."""
    specs = [(3, 90, 20), (5, 150, 35), (11, 60, 12), (12, 30, 8), (13, 120, 25), (14, 40, 30), (15, 100, 18), (16, 70, 28)]
    chunks = [synth.make_ont_chunk(seed=s, region_bp=n * 500, n_sites=n, coverage=c) for s, n, c in specs]
    empty = synth.make_ont_chunk(seed=99, region_bp=10_000, n_sites=20, coverage=5)
    empty.reads = []
    chunks.insert(4, empty)
    pd = synth.shipped_phase_params()
    params = capi.Params.from_reference_names(pd)
    q = capi.Queue([0, 0])
    try:
        got, st = q.phase(chunks, params, chunks_per_batch=3)
        assert st.n_devices == 2 and st.batches == 3 and sum(st.chunks_per_device[:2]) == len(chunks)
        assert sum(st.units_per_device[:2]) == sum(c.units for c in chunks) and st.fallback_chunks == 0
        for chunk, g in zip(chunks, got):
            if not chunk.reads:
                assert g["length"] == 0 and g["reads1"] == [] and g["reads2"] == []
                continue
            oc = orc.OracleChunk(chunk)
            ref = oc.phase(pd)
            oc.close()
            for k in ("hap1", "hap2", "genotype", "ancestor", "support1", "support2", "genotype_probs", "hap_probs1", "hap_probs2"):
                assert (np.asarray(g[k]) == np.asarray(ref[k])).all(), k
            assert g["reads1"] == ref["reads1"] and g["reads2"] == ref["reads2"]
        again, st2 = q.phase(chunks, params, chunks_per_batch=5)
        assert st2.batches == 2
        for a, b in zip(got, again):
            assert (np.asarray(a["hap1"]) == np.asarray(b["hap1"])).all() and a["reads1"] == b["reads1"]
        # more batches than pulling threads (two devices x four lanes): every lane stages its next batch beside the current call
        twice = chunks + chunks
        many, st3 = q.phase(twice, params, chunks_per_batch=1)
        assert st3.batches == len(twice) and sum(st3.chunks_per_device[:2]) == len(twice)
        for a, b in zip(got + got, many):
            assert (np.asarray(a["hap1"]) == np.asarray(b["hap1"])).all() and (np.asarray(a["hap2"]) == np.asarray(b["hap2"])).all()
            assert a["reads1"] == b["reads1"] and a["reads2"] == b["reads2"]
    finally:
        q.close()
    # the one-shot form, one worker
    one, st1 = capi.phase_chunks_on_devices([0], chunks[:3], params, chunks_per_batch=2)
    assert st1.batches == 2 and all((np.asarray(a["hap1"]) == np.asarray(b["hap1"])).all() for a, b in zip(one, got[:3]))
    with pytest.raises(capi.MrpError) as ei:
        capi.phase_chunks_on_devices([7], chunks[:1], params)
    assert ei.value.code == capi.MRP_ERR_ARG


def test_library_batches_are_cut_by_units_for_mixed_chunk_sizes():
    """chunks_per_batch = 0 on one device: 1 200 chunks of the 1 Mb kind (~60 000 units) mixed with 6 000 of the 100 kb kind (~4 000
    units) and a few without reads.  More than 1 280 yardstick chunks of work, so the queue is cut into batches of about
    192 x 60 000 units each, in cost order: the first batches hold ~190 large chunks, the last ones thousands of small ones; every
    chunk is in exactly one batch."""
    rng = np.random.default_rng(5)
    cost = np.concatenate([rng.integers(50_000, 70_000, size=1200), rng.integers(2_000, 6_000, size=6_000), np.zeros(7, dtype=np.int64)]).astype(np.int64)
    rng.shuffle(cost)
    order, batch = capi.queue_plan(cost, 0)
    assert sorted(order.tolist()) == list(range(len(cost)))
    assert (np.diff(cost[order]) <= 0).all() and (np.diff(batch[order]) >= 0).all() and batch[order][0] == 0
    per_batch = np.bincount(batch, weights=cost.astype(np.float64))
    n_batches = int(np.ceil(cost.sum() / (192 * 60_000)))   # whole rounds of the (here: one) lane, equal batches of at most 192 x 60 000 units
    target = cost.sum() / n_batches
    assert len(per_batch) == n_batches
    assert (np.abs(per_batch[:-1] / target - 1.0) < 0.01).all() and 0.5 * target <= per_batch[-1] <= 1.5 * target
    sizes = np.bincount(batch)
    assert 140 <= sizes[0] <= 230 and sizes[-1] > 1000
