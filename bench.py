#!/usr/bin/env python3
"""Benchmark of the stRPHmm forward/backward hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic 1 Mb chunks, 2 000 biallelic het
sites, 30x ONT-like reads, shipped ONT haplotag parameters (max-plus mode, 100 partitions per
column).  One *step* = every forward/backward sweep needed to phase a batch of such chunks (all
overlap components of every tiling-path merge level + the final sweep of each chunk), with the
flattened HMMs already resident in HBM: bit-plane kernel -> emission kernel -> recursion kernel.
A single 2 000-column HMM is a strictly sequential chain, so the chip is filled by keeping the
sweeps of many independent chunks in flight (the reference's own parallel axis, phase.c:276).

The job set is produced by the product's host pipeline (margin_amd/csrc/rphmm_host.c) running the
real merge recursion with device sweeps; it is recorded into one device batch and replayed in the
timed region.  The oracle is used only by the cpu_baseline leg.

Prints ONE JSON line on rank 0 (see the contract in the task description).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chunks", type=int, default=int(os.environ.get("MRP_BENCH_CHUNKS", "96")),
                    help="synthetic 1 Mb chunks resident per GPU")
    ap.add_argument("--sites", type=int, default=2000)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--threads", type=int, default=int(os.environ.get("MRP_BENCH_THREADS", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the end-to-end device-resident phasing leg")
    ap.add_argument("--pipeline-runs", type=int, default=3)
    ap.add_argument("--align-chunks", type=int, default=4, help="chunks whose read x allele pairs the alignment leg scores (0: skip)")
    ap.add_argument("--align-runs", type=int, default=3)
    ap.add_argument("--pipeline-groups", type=int, default=1,
                    help="additional caller-side split of the chunks into concurrent mrp_phase_reads_many calls (the call itself "
                         "already runs two interleaved halves on sibling contexts, MRP_PHASE_GROUPS)")
    ap.add_argument("--split", type=int, default=int(os.environ.get("MRP_BENCH_SPLIT", "1")),
                    help="record the chunks into this many device batches launched on separate streams (their kernels overlap)")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        # rehearsal knobs (not used by the driver): several ranks may share one card with gloo
        backend = os.environ.get("MRP_BENCH_BACKEND", "nccl")
        if "MRP_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["MRP_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    import numpy as np
    from margin_amd import capi, sharding, synth

    params_dict = synth.shipped_phase_params()
    params = capi.Params.from_reference_names(params_dict)
    n_chunks = args.chunks
    cpu_share = max(1, (os.cpu_count() or 8) // max(1, world if world > 1 else 1))
    n_threads = args.threads or min(16, cpu_share, n_chunks)

    ctxs = [capi.Context(local_rank) for _ in range(max(1, args.split))]
    bigs = [capi.Batch(c) for c in ctxs]
    main_ctx, big = ctxs[0], bigs[0]
    keep = []           # device chunks must outlive the batch
    host_chunks = [None] * n_chunks
    units_lock = threading.Lock()
    totals = dict(units=0, sweeps=0, reads=0)
    tls = threading.local()

    seeds = sharding.chunk_seeds(rank, n_chunks)

    def build_one(i):
        if not hasattr(tls, "ctx"):
            tls.ctx = capi.Context(local_rank)  # one context (stream) per host thread
        seed = seeds[i]
        chunk = synth.make_ont_chunk(seed=seed, region_bp=args.sites * 500, n_sites=args.sites,
                                     coverage=args.coverage)
        dchunk = capi.DeviceChunk.from_chunk(tls.ctx, chunk)
        res = capi.phase_reads(tls.ctx, dchunk, chunk, params, record=bigs[i % len(bigs)])
        with units_lock:
            keep.append(dchunk)
            totals["units"] += chunk.units
            totals["sweeps"] += res["n_sweeps"]
            totals["reads"] += len(chunk.reads)
        host_chunks[i] = chunk
        return chunk if i == 0 else None

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=n_threads) as ex:
        first = list(ex.map(build_one, range(n_chunks)))[0]
    t_build = time.time() - t0
    for b_ in bigs:
        b_.upload()
    for c_ in ctxs:
        c_.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        for b_ in bigs:
            b_.launch()
    for c_ in ctxs:
        c_.synchronize()
    for b_ in bigs:
        b_.stats()  # closes the averaging window of the warm-up launches
    barrier()
    # The K steps are queued back to back: every launch keeps its own HIP events inside the library (mrp_launch_stats
    # avg_*: the kernels' durations averaged over the launches of the timed region), so nothing waits on the host between
    # steps (with MRP_PRE_STREAM=1 the byte packing of step k + 1 then runs beside the recursion kernels of step k).
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for b_ in bigs:
            b_.launch()
    for c_ in ctxs:
        c_.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    s = bigs[-1].stats()
    assert s.launches_averaged == min(args.steps, 32)  # the library keeps the events of a batch's 32 most recent launches
    planes_ms, emission_ms, sweep_ms = [s.avg_planes_ms], [s.avg_emission_ms], [s.avg_sweep_ms]
    reduce_dev = "cuda" if (dist is not None and dist.get_backend() == "nccl") else None
    elapsed, units_all = sharding.reduce_elapsed_and_units(dist, elapsed, float(totals["units"]), device=reduce_dev)

    sts = [b_.stats() for b_ in bigs]
    st = sts[0]
    for o_ in sts[1:]:
        for f_ in ("n_hmms", "n_columns", "n_cells", "n_merge_cells", "profile_bytes", "algorithmic_bytes", "popcount_ops"):
            setattr(st, f_, getattr(st, f_) + getattr(o_, f_))
    value = units_all * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    sweep_avg = float(np.mean(sweep_ms))
    emis_avg = float(np.mean(emission_ms))
    planes_avg = float(np.mean(planes_ms))
    alg = float(st.algorithmic_bytes)
    # B = sum_k 24*C_k + 32*M_k + D_k*Al_k + 8 for the whole sweep.  The sweep is three
    # kernels here; each is credited with the algorithmic bytes it is responsible for:
    #   emission kernel : partition read 8*C (+ the profile bytes D*Al via the plane kernel)
    #   recursion kernel: f and b 16*C, merge cells 32*M, column totals 8*K      <- dominant kernel
    C, M, K = float(st.n_cells), float(st.n_merge_cells), float(st.n_columns)
    alg_sweep = 16.0 * C + 32.0 * M + 8.0 * K
    alg_emission = 8.0 * C
    achieved = alg_sweep / (sweep_avg * 1e-3) / 1e9
    whole = alg / (ms_per_step * 1e-3) / 1e9
    # HBM bytes the kernels actually move per launch (by construction; PMC cross-check in profiles/):
    #   emission 8*C read + 4*C write; recursion 2 * (8*C read + 4*C write) + 2 * 4*M write
    moved = 12.0 * C + 24.0 * C + 8.0 * M
    roofline = dict(bound="hbm", kernel="mrp_sweep_i32_kernel", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=None,
                    # PMC counters cannot be read from inside this process; measured with rocprofv3 (separate --pmc passes,
                    # FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes) on the same command at 32 chunks:
                    traffic_profile=dict(source="profiles/r01/replay_32chunks_v3_summary.txt",
                                         hbm_bytes_per_algorithmic_byte=0.79,
                                         note="recursion kernel, 3 size classes: 2 x FETCH_SIZE 2.68 GB + WRITE_SIZE 4.52 GB = 9.88 GB per step "
                                              "against 12.5 GB algorithmic (results are stored as int32, the reference's formula counts doubles)"),
                    algorithmic_bytes_per_launch=alg_sweep, kernel_ms=sweep_avg,
                    whole_step=dict(achieved=whole, frac=whole / HBM_PEAK_GBS, algorithmic_bytes=alg,
                                    moved_bytes_model=moved, moved_GBps=moved / (ms_per_step * 1e-3) / 1e9,
                                    planes_ms=planes_avg, emission_ms=emis_avg, sweep_ms=sweep_avg,
                                    emission_kernel=dict(algorithmic_bytes=alg_emission,
                                                         achieved=alg_emission / (emis_avg * 1e-3) / 1e9)),
                    popcount64_per_s=float(st.popcount_ops) / (ms_per_step * 1e-3))

    out = dict(metric="het-sites x reads phased/sec (stRPHmm forward/backward sweeps, 30x ONT synthetic)",
               value=value, unit="het-site-reads/s", n_gpus=n_gpus, steps=args.steps, warmup=args.warmup,
               ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="int32",
               data="synthetic",
               config=dict(workload=f"configs[1]: synthetic 1 Mb chunk, {args.sites} het sites, {args.coverage:g}x ONT reads, "
                                    f"shipped ONT haplotag params; {n_chunks} chunks resident per GPU, all merge-level + final sweeps",
                           chunks_per_gpu=n_chunks, hmms_per_gpu=int(st.n_hmms), columns_per_gpu=int(st.n_columns),
                           cells_per_gpu=int(st.n_cells), merge_cells_per_gpu=int(st.n_merge_cells),
                           units_per_gpu=int(totals["units"]), sweeps_per_gpu=int(totals["sweeps"]),
                           parallelism=f"{world} process(es), one per GPU, chunks sharded, no collectives",
                           host_build_s=t_build, host_threads=n_threads),
               roofline=roofline)

    if not args.no_pipeline:
        # End-to-end leg: the same chunks phased from profile sequences to haplotypes by
        # mrp_phase_reads_many -- tiling paths, every merge level (cross product -> forward/backward -> prune, resident
        # in HBM), fused final sweep, trace back, genome fragments.  Wall clock around the C call, inputs (profile bytes,
        # site tables) already on the device; it is reported beside the headline value, not as it.
        G = max(1, min(args.pipeline_groups, n_chunks))
        gctx = [capi.Context(local_rank) for _ in range(G)]
        gchunks = [host_chunks[g::G] for g in range(G)]
        pdch = [[capi.DeviceChunk.from_chunk(gctx[g], c) for c in gchunks[g]] for g in range(G)]
        for c in host_chunks:
            capi.read_records(c)

        def run_groups():
            with ThreadPoolExecutor(max_workers=G) as ex:
                return list(ex.map(lambda g: capi.phase_reads_many(gctx[g], pdch[g], gchunks[g], params, convert=False)[1], range(G)))

        run_groups()  # warm-up: allocator cache, pinned buffers
        barrier()
        t0 = time.perf_counter()
        psts = None
        for _ in range(args.pipeline_runs):
            psts = run_groups()
        barrier()
        p_el = time.perf_counter() - t0
        p_el, p_units = sharding.reduce_elapsed_and_units(dist, p_el, float(totals["units"]), device=reduce_dev)
        out["pipeline"] = dict(what="mrp_phase_reads_many: profile sequences -> haplotypes, all merge levels resident on the device",
                               value=p_units * args.pipeline_runs / p_el, unit="het-site-reads/s",
                               ms_per_batch=1e3 * p_el / args.pipeline_runs, chunks_per_gpu=n_chunks, runs=args.pipeline_runs,
                               concurrent_batches=G * int(os.environ.get("MRP_PHASE_GROUPS", "2")), resident=int(all(p.resident for p in psts)), levels=int(psts[0].levels),
                               hmms=int(sum(p.hmms for p in psts)), columns=int(sum(p.columns for p in psts)),
                               cells=int(sum(p.cells for p in psts)), device_ms=float(sum(p.device_ms for p in psts)),
                               cross_ms=float(sum(p.cross_ms for p in psts)), sweep_ms=float(sum(p.sweep_ms for p in psts)),
                               prune_ms=float(sum(p.prune_ms for p in psts)),
                               host_threads=int(os.environ.get("MRP_HOST_THREADS", "0")) or min(16, os.cpu_count() or 1))
        for grp in pdch:
            for d_ in grp:
                d_.close()
        for c_ in gctx:
            c_.close()

    if args.align_chunks > 0:
        # Alignment leg: the banded pair-HMM forward probability of every read substring against every
        # allele of every site (bubbleGraph.c:1421-1464), the numbers that become the profile bytes the sweep above reads.
        # value = pairs / kernel time (HIP events inside the library, strings already on the device); the rate of the
        # whole call (classification, upload, download) is given beside it.
        t_hmm, t_tr, t_em = synth.margin_phase_pair_hmm_arrays()
        fwd = capi.PairHmm.from_margin_hmm(t_hmm, t_tr, t_em)
        a_models = [fwd, fwd.reverse_complement()]
        bubbles = []
        for c in range(args.align_chunks):
            bubbles += synth.make_bubble_strings(seed=1000 * rank + c + 1, n_sites=args.sites, coverage=int(args.coverage))
        a_in = synth.pairs_from_bubbles(bubbles)
        actx = capi.Context(local_rank)
        capi.forward_probabilities(actx, a_models, *a_in)  # warm-up
        barrier()
        t0 = time.perf_counter()
        k_ms, cells = 0.0, 0
        for _ in range(args.align_runs):
            a_out, a_st = capi.forward_probabilities(actx, a_models, *a_in)
            k_ms += a_st.kernel_ms
            cells = a_st.cells
        barrier()
        a_el = time.perf_counter() - t0
        n_pairs = len(a_in[1])
        k_s, _ = sharding.reduce_elapsed_and_units(dist, k_ms / 1e3, float(n_pairs), device=reduce_dev)
        a_el, a_pairs = sharding.reduce_elapsed_and_units(dist, a_el, float(n_pairs), device=reduce_dev)
        _, cells_all = sharding.reduce_elapsed_and_units(dist, 0.0, float(cells), device=reduce_dev)
        out["alignment"] = dict(what="mrp_forward_probabilities: banded pair-HMM forward log probability, read substring x allele",
                                value=a_pairs * args.align_runs / k_s, unit="pairs/s", cells_per_s=cells_all * args.align_runs / k_s,
                                kernel_ms=1e3 * k_s / args.align_runs, call_value=a_pairs * args.align_runs / a_el,
                                call_ms=1e3 * a_el / args.align_runs, pairs_per_gpu=n_pairs, cells_per_gpu=int(cells), chunks_per_gpu=args.align_chunks,
                                dtype="f64", parity="bit-exact vs oracle/pairhmm_oracle.c (no fused multiply-add on either side)",
                                bound="fp64 VALU issue", runs=args.align_runs)
        if rank == 0 and not args.no_cpu_baseline and n_gpus == 1:
            from oracle import pairhmm as ph
            om = [ph.Model.from_buffer_copy(bytes(m)) for m in a_models]
            n_thr = max(1, min(16, os.cpu_count() or 1))
            per = 8000
            pool_, xo, xl, yo, yl, mi = a_in
            parts = [slice(t * per, (t + 1) * per) for t in range(n_thr) if t * per < n_pairs]

            def cpu_align(sl):
                t1 = time.perf_counter()
                r = ph.forward_batch(om, pool_, xo[sl], xl[sl], yo[sl], yl[sl], mi[sl])
                return time.perf_counter() - t1, r

            with ThreadPoolExecutor(max_workers=len(parts)) as ex:
                res_a = list(ex.map(cpu_align, parts))
            same = all((r[1] == a_out[sl]).all() for r, sl in zip(res_a, parts))
            n_s = sum(len(r[1]) for r in res_a)
            out["alignment"]["cpu_baseline"] = dict(value=n_s / max(r[0] for r in res_a), unit="pairs/s", cores=len(parts), kind="port",
                                                    sample=f"{n_s} of {n_pairs} pairs, {per} per thread", per_core=len(res_a[0][1]) / res_a[0][0],
                                                    identical_to_gpu=bool(same))
        actx.close()

    if rank == 0 and not args.no_cpu_baseline and n_gpus == 1:
        # CPU baseline: the oracle (C restatement of the reference's linked-list/hash implementation, -O3 -mpopcnt)
        # phasing one chunk per thread -- the reference's own parallel axis (phase.c:276) -- on a bounded sample of
        # the same workload; only the time inside its stRPHmm_forwardBackward calls is counted, like the GPU value.
        from oracle import orc
        n_thr = max(1, min(16, os.cpu_count() or 1, n_chunks))

        def cpu_one(c):
            oc = orc.OracleChunk(c)
            r = oc.phase(params_dict)
            oc.close()
            return r["fb_seconds"], r["fb_calls"]

        t_cpu = time.perf_counter()
        with ThreadPoolExecutor(max_workers=n_thr) as ex:
            res = list(ex.map(cpu_one, host_chunks[:n_thr]))
        t_cpu = time.perf_counter() - t_cpu
        sample_units = sum(c.units for c in host_chunks[:n_thr])
        slowest = max(r[0] for r in res)
        out["cpu_baseline"] = dict(value=sample_units / slowest, unit="het-site-reads/s", cores=n_thr, kind="port",
                                   sample=f"{n_thr} of {n_chunks} chunks, one per thread: all {sum(r[1] for r in res)} sweeps "
                                          f"({sample_units} units); slowest thread spent {slowest:.2f} s in forward/backward "
                                          f"(whole phasing of the sample: {t_cpu:.1f} s wall)",
                                   per_core=first.units / res[0][0])
    if rank == 0:
        print(json.dumps(out))
    for b_ in bigs:
        b_.close()
    for d in keep:
        d.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
