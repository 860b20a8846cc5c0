#!/usr/bin/env python3
"""Benchmark of the stRPHmm forward/backward hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic 1 Mb chunks, 2 000 biallelic het sites, 30x ONT-like
reads, shipped ONT haplotag parameters (max-plus mode, 100 partitions per column).

One *step* = phasing every chunk of the GPU once, end to end, through the product's C-ABI (mrp_phase_reads_many):
tiling paths, every merge level (cross product -> forward/backward -> prune, resident in HBM), the final sweep with the
ancestor model, trace back, genome fragments and the read bipartition -- everything bubbleGraph_phaseBubbleGraph does from
profile sequences on.  Inputs (profile bytes, site tables) are resident in HBM when the timed region starts.
`value` = het-sites x reads of all chunks x steps / wall time: the rate at which chunks are actually phased.

Beside it, as evidence for the kernels (not as the headline):
  roofline      the forward/backward recursion kernel (the kernel that moves the algorithmic bytes of SURVEY.md 8d) replayed
                over ALL sweeps of the same chunks as one batch: HIP-event duration per launch against its algorithmic bytes;
                `traffic` = HBM bytes of that launch from the committed rocprofv3 PMC pass (profiles/r02/traffic.json);
                `path` = the algorithmic bytes of all sweeps over the wall time of the real step.
  queue         the same chunks through the host work queue from HOST memory (upload included, PCIe-inclusive rate)
  shapes        the chunk shapes of BASELINE.json configs[2] (640 x ~130 sites) and configs[4] (HiFi, 2-4 alleles) through the same call
  alignment     the pair-HMM kernel family that produces the profile bytes
  cpu_baseline  the oracle (CPU restatement of the reference) on a bounded sample of the same chunks, N = 1 only

Launch: `python bench.py` (one GPU), or one rank per GPU under torch.distributed.run with --gpus N = WORLD_SIZE (chunks
are sharded by rank, no collective on the data path), or `python bench.py --gpus N` in ONE process: the library's work
queue (mrp_queue_*) then drives N devices itself.  Any other combination is refused.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=int(os.environ.get("MRP_BENCH_CHUNKS", "1152")), help="synthetic 1 Mb chunks per GPU")
    ap.add_argument("--sites", type=int, default=2000)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--threads", type=int, default=int(os.environ.get("MRP_BENCH_THREADS", "0")))
    ap.add_argument("--phase-groups", type=int, default=0,
                    help="concurrent batches inside mrp_phase_reads_many (0: the library's choice, one per 24 chunks up to 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the kernel replay leg (it needs ~20 s of host work to record the sweeps)")
    ap.add_argument("--roofline-steps", type=int, default=20)
    ap.add_argument("--roofline-chunks", type=int, default=96, help="chunks whose sweeps the kernel replay leg records (host work: ~0.2 s per chunk)")
    ap.add_argument("--queue-runs", type=int, default=3, help="runs of the host-memory work queue leg (0: skip)")
    ap.add_argument("--queue-batch", type=int, default=0, help="chunks per batch of the queue leg (0: the library's default, 96)")
    ap.add_argument("--shape-runs", type=int, default=3, help="runs of the configs[2] / configs[4] shape legs (0: skip)")
    ap.add_argument("--align-chunks", type=int, default=4, help="chunks whose read x allele pairs the alignment leg scores (0: skip)")
    ap.add_argument("--align-runs", type=int, default=3)
    ap.add_argument("--sum-chunks", type=int, default=16, help="chunks whose sweeps the log-sum-exp leg replays (0: skip)")
    ap.add_argument("--queue-child", action="store_true", help="(internal) run the work queue leg alone and print its runs as one JSON line")
    return ap.parse_args()


def queue_child(args):
    """The work queue leg in a process of its own (started by the parent before IT touches the GPU): a caller of the queue holds no
    other family of contexts, and the queue's workers bring contexts, streams, a host pool and allocator caches of their own --
    whichever of the two families of a process comes second runs 3-6 % slower at 1 152 chunks (measured both ways round)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    from margin_amd import capi, sharding, synth
    params = capi.Params.from_reference_names(synth.shipped_phase_params())
    cpu_share = max(1, os.cpu_count() or 8)
    capi.load().mrp_set_host_threads(max(1, min(16, cpu_share)))
    n_threads = args.threads or min(16, cpu_share, args.chunks)
    seeds = sharding.chunk_seeds(0, args.chunks)
    with ThreadPoolExecutor(max_workers=n_threads) as ex:
        chunks = list(ex.map(lambda sd: synth.make_ont_chunk(seed=sd, region_bp=args.sites * 500, n_sites=args.sites, coverage=args.coverage), seeds))
    for c in chunks:
        capi.read_records(c)
    descs = capi.chunk_descs(chunks)
    q = capi.Queue([0])
    for _ in range(2):
        q.phase(chunks, params, chunks_per_batch=args.queue_batch, descs=descs, convert=False)
    runs = []
    for _ in range(args.queue_runs):
        t0 = time.perf_counter()
        _, qst = q.phase(chunks, params, chunks_per_batch=args.queue_batch, descs=descs, convert=False)
        runs.append(1e3 * (time.perf_counter() - t0))
    q.close()
    print(json.dumps(dict(queue_child=True, runs_ms=runs, batches=int(qst.batches), units=float(sum(c.units for c in chunks)))))


def run_queue_child(args):
    """parent side: None if the child could not be run (the in-process leg is then the only one)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--queue-child", "--chunks", str(args.chunks), "--sites", str(args.sites),
           "--coverage", str(args.coverage), "--queue-runs", str(args.queue_runs), "--queue-batch", str(args.queue_batch), "--threads", str(args.threads)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        for line in reversed(res.stdout.splitlines()):
            if line.startswith("{") and "queue_child" in line:
                return json.loads(line)
    except Exception:
        pass
    return None


def insitu_rooflines(st):
    """Per kernel family IN THE REAL STEP (not the replay): the algorithmic bytes the family is responsible for (SURVEY.md 8d:
    recursion 16 B per cell + 32 B per merge cell + 8 B per column; prune: the 8 B per cell of f and b it ranks; cross product +
    emission: the 8 B per cell of cost and transitions it writes) over the family's SUMMED device time of the step.  The
    concurrent batches of a call run side by side, so the summed time exceeds the wall time and the rate is what one batch's
    kernels get while sharing the device: roofline.path is the product of these terms and of the overlap."""
    C, M, K = float(st.cells), float(st.merge_cells), float(st.columns)
    fam = {"recursion": (16.0 * C + 32.0 * M + 8.0 * K, float(st.recursion_ms), "mrp_sweep_i32_kernel"),
           "prune": (8.0 * C, float(st.prune_kernel_ms), "mrp_prune_kernel"),
           "cross_product_emission": (8.0 * C, float(st.cross_emit_ms), "mrp_cross_emit_kernel (+ the final level's cross / emission kernels)")}
    out = {}
    for name, (b, ms, kern) in fam.items():
        if ms > 0:
            out[name] = dict(kernel=kern, algorithmic_bytes=b, summed_device_ms=ms, achieved=b / (ms * 1e-3) / 1e9, unit="GB/s",
                             frac=b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
    return out


def bind_near_device(device):
    """sched_setaffinity to /sys/bus/pci/devices/<bus id of the device>/local_cpulist (intersected with the CPUs the process may
    use); returns what was done, for the JSON line"""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 63, int(device)) != 0:
            return "none (no PCI bus id)"
        bus = buf.value.decode().lower()
        with open(f"/sys/bus/pci/devices/{bus}/local_cpulist") as f:
            text = f.read().strip()
        near = set()
        for tok in text.split(","):
            if "-" in tok:
                lo, hi = tok.split("-")
                near.update(range(int(lo), int(hi) + 1))
            elif tok:
                near.add(int(tok))
        allowed = os.sched_getaffinity(0)
        want = near & allowed
        if len(want) < 8 or want == allowed:
            return f"none (local_cpulist of {bus} leaves {len(want)} of the {len(allowed)} usable CPUs)"
        os.sched_setaffinity(0, want)
        return f"{len(want)} CPUs next to {bus}"
    except Exception as e:  # no sysfs entry, a container without the call, ...
        return f"none ({type(e).__name__})"


def main():
    args = parse_args()
    if args.queue_child:
        return queue_child(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    # the work queue leg in a fresh process, BEFORE this one touches the GPU (a process that has initialised the GPU must not start
    # another program); one rank only: the ranks of a multi-GPU launch keep the in-process leg, whose runs they can bracket by barriers
    fresh_queue = None
    if world == 1 and n_gpus == 1 and args.queue_runs > 0 and args.steps > 0 and "MRP_BENCH_DEVICES" not in os.environ:
        fresh_queue = run_queue_child(args)
    # one rank per GPU (the driver's launch), or one process that drives the N devices through the library's queue
    if world != 1 and world != n_gpus:
        sys.exit(f"bench.py: --gpus {n_gpus} but WORLD_SIZE={world}: launch one rank per GPU (torch.distributed.run --nproc-per-node {n_gpus}) "
                 f"or a single process")
    single_process_multi = world == 1 and n_gpus > 1

    # torch initialises HIP before libmargin_rphmm.so is loaded: the library's own default (its concurrent batches need more
    # hardware queues than the runtime's 4, include/margin_rphmm.h) has to be in the environment by then
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        backend = os.environ.get("MRP_BENCH_BACKEND", "nccl")  # rehearsal knob: gloo with several ranks on one card
        if "MRP_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["MRP_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    # one rank per GPU on a node with several sockets: the rank's threads (this process's, the library's pool, the runtime's) stay on the
    # CPUs next to its device, as the library's work queue does for its workers -- staging buffers and the hmm shadows are then local
    # memory.  Only inside what the process may use anyway, and never down to a handful of CPUs.
    cpu_binding = "none"
    if world > 1 and os.environ.get("MRP_BENCH_AFFINITY", "1") != "0":
        cpu_binding = bind_near_device(local_rank)

    import numpy as np
    from margin_amd import capi, sharding, synth

    # rehearsal knob: the single-process path over an explicit device list (e.g. "0,0": two queue workers sharing one card)
    queue_devices = [int(x) for x in os.environ["MRP_BENCH_DEVICES"].split(",")] if os.environ.get("MRP_BENCH_DEVICES") else list(range(n_gpus))
    if single_process_multi and (len(queue_devices) != n_gpus or capi.load().mrp_device_count() <= max(queue_devices)):
        sys.exit(f"bench.py: --gpus {n_gpus} in one process over devices {queue_devices}, but {capi.load().mrp_device_count()} device(s) are visible")

    params_dict = synth.shipped_phase_params()
    params = capi.Params.from_reference_names(params_dict)
    n_chunks = args.chunks * (n_gpus if single_process_multi else 1)
    cpu_share = max(1, (os.cpu_count() or 8) // max(1, world))
    n_threads = args.threads or min(16, cpu_share, n_chunks)
    # the library's worker pool: this rank's share of the node's cores, at most 16 per device (a work queue gives every
    # device its own pool of this size)
    host_threads = max(1, min(16, cpu_share // (n_gpus if single_process_multi else 1)))
    capi.load().mrp_set_host_threads(host_threads)
    seeds = sharding.chunk_seeds(rank, n_chunks)

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=n_threads) as ex:
        chunks = list(ex.map(lambda s: synth.make_ont_chunk(seed=s, region_bp=args.sites * 500, n_sites=args.sites, coverage=args.coverage), seeds))
    t_synth = time.time() - t0
    units = float(sum(c.units for c in chunks))
    for c in chunks:
        capi.read_records(c)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    reduce_dev = "cuda" if (dist is not None and dist.get_backend() == "nccl") else None

    # ---- the timed region: every chunk phased end to end, K times -------------------------------------------------
    if single_process_multi:
        queue = capi.Queue(queue_devices)
        descs = capi.chunk_descs(chunks)
        step = lambda: queue.phase(chunks, params, chunks_per_batch=args.queue_batch, descs=descs, convert=False)[1]
    else:
        ctx = capi.Context(local_rank)
        ctx.set_phase_groups(args.phase_groups)
        dchunks = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
        many_args = capi.phase_many_args(dchunks, chunks)  # (the call's argument arrays: built once, not per step)
        step = lambda: capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=many_args)[1]
    for _ in range(args.warmup):
        step()
    barrier()
    t0, c0 = time.perf_counter(), time.process_time()
    st = None
    for _ in range(args.steps):
        st = step()
    barrier()
    elapsed, host_cpu = time.perf_counter() - t0, time.process_time() - c0
    elapsed, units_all = sharding.reduce_elapsed_and_units(dist, elapsed, units, device=reduce_dev)
    # (--steps 0: the counter passes of tools/collect_profiles_r04.sh, which want the replay leg's launches alone in the trace)
    value = units_all * args.steps / elapsed if args.steps > 0 else 0.0
    ms_per_step = 1e3 * elapsed / args.steps if args.steps > 0 else float("nan")

    cfg = dict(workload=f"configs[1]: synthetic 1 Mb chunk, {args.sites} het sites, {args.coverage:g}x ONT reads, shipped ONT haplotag params; "
                        f"{args.chunks} chunks per GPU phased end to end per step (mrp_phase_reads_many: all merge levels resident in HBM, final sweep, "
                        f"trace back, genome fragments)",
               chunks_per_gpu=args.chunks, units_per_gpu=int(units) // (n_gpus if single_process_multi else 1),
               parallelism=(f"1 process, {n_gpus} devices, host work queue (mrp_queue_phase_chunks), no collectives" if single_process_multi else
                            f"{world} process(es), one per GPU, chunks sharded by rank, no collectives"),
               host_threads=host_threads, host_cpu_s_per_step=host_cpu / max(1, args.steps), synth_s=t_synth, cpu_binding=cpu_binding)
    out = dict(metric="het-sites x reads phased/sec (30x ONT synthetic chunks, end to end: every merge level + final sweep)",
               value=value, unit="het-site-reads/s", n_gpus=n_gpus, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="int32", data="synthetic", config=cfg)
    if not single_process_multi and st is not None:
        out["step_detail"] = dict(resident=int(st.resident), fallback_chunks=int(st.fallback_chunks), levels=int(st.levels), hmms=int(st.hmms),
                                  columns=int(st.columns), cells=int(st.cells), merge_cells=int(st.merge_cells), device_ms=float(st.device_ms),
                                  cross_ms=float(st.cross_ms), sweep_ms=float(st.sweep_ms), prune_ms=float(st.prune_ms),
                                  pack_ms=float(st.pack_ms), cross_emit_ms=float(st.cross_emit_ms), recursion_ms=float(st.recursion_ms),
                                  prune_kernel_ms=float(st.prune_kernel_ms), compact_ms=float(st.compact_ms),
                                  note="device_ms: summed HIP-event time of the levels' kernels of the last step (the concurrent batches of a call add up: "
                                       "their kernels share the device); cross_ms / sweep_ms / prune_ms are the level's three event intervals (sweep_ms includes "
                                       "packing and the one-pass cross product + emission, prune_ms the compaction), the *_ms after them the same time by kernel family")
        out["insitu"] = insitu_rooflines(st)
    elif st is not None:
        out["step_detail"] = dict(batches=int(st.batches), fallback_chunks=int(st.fallback_chunks),
                                  chunks_per_device=[int(st.chunks_per_device[d]) for d in range(n_gpus)],
                                  busy_ms_per_device=[float(st.busy_ms_per_device[d]) for d in range(n_gpus)],
                                  note="inputs in host memory: the upload of every batch is inside the timed region")

    # ---- latency: what ONE chunk and EIGHT chunks of the headline workload cost end to end (BASELINE.json configs[1] is literally
    # "a single 1 Mb chunk, one sweep": a caller with few chunks in hand sees this, not the many-chunk rate) ----------------
    if not single_process_multi and rank == 0:
        lat = {}
        for m in (1, 8):
            if m > len(dchunks):
                continue
            capi.phase_reads_many(ctx, dchunks[:m], chunks[:m], params, convert=False)
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                capi.phase_reads_many(ctx, dchunks[:m], chunks[:m], params, convert=False)
            dt = (time.perf_counter() - t1) / reps
            lat[f"{m}_chunk{'s' if m > 1 else ''}"] = dict(ms_per_call=1e3 * dt, value=float(sum(c.units for c in chunks[:m])) / dt, unit="het-site-reads/s")
        out["latency"] = dict(what="one mrp_phase_reads_many call over 1 and over 8 chunks of the headline workload, inputs resident in HBM, end to end", **lat)
    if dist is not None:
        dist.barrier()

    # ---- the same chunks from HOST memory through the work queue (PCIe-inclusive) --------------------------------
    if args.queue_runs > 0 and not single_process_multi:
        # the queue's workers have contexts, streams and allocator caches of their own: the resident leg's go first (a caller of
        # the queue holds no second family of contexts whose streams would share the device's hardware queues with the queue's)
        for d_ in dchunks:
            d_.close()
        ctx.close()
        q = capi.Queue([local_rank])
        descs = capi.chunk_descs(chunks)
        q.phase(chunks, params, chunks_per_batch=args.queue_batch, descs=descs, convert=False)
        barrier()
        q_ms = []
        for _ in range(args.queue_runs):
            t0 = time.perf_counter()
            _, qst = q.phase(chunks, params, chunks_per_batch=args.queue_batch, descs=descs, convert=False)
            barrier()
            q_ms.append(1e3 * (time.perf_counter() - t0))
        # the median run (every run is listed): a run that meets a cold allocator after another tenant's memory pressure takes seconds
        q_el = sorted(q_ms)[len(q_ms) // 2] * 1e-3
        q_el, q_units = sharding.reduce_elapsed_and_units(dist, q_el, units, device=reduce_dev)
        out["queue"] = dict(what="mrp_queue_phase_chunks: the same chunks from HOST memory (PCIe-inclusive): sorted by estimated cost, pulled in "
                                 "batches by one worker per device, the next batch's site tables and profile bytes uploaded on a second stream "
                                 "while the current batch is phased",
                            value=q_units / q_el, unit="het-site-reads/s", ms_per_run=1e3 * q_el, runs_ms=[round(x, 1) for x in q_ms],
                            batches=int(qst.batches), runs=args.queue_runs, vs_resident=(q_units / q_el) / value)
        if fresh_queue is not None and fresh_queue.get("runs_ms"):
            # the same leg from a process of its own (run_queue_child): that is what a caller of the queue sees; the in-process runs above
            # come second in a process whose first family of contexts -- closed by now -- was the resident leg's
            f_ms = sorted(fresh_queue["runs_ms"])[len(fresh_queue["runs_ms"]) // 2]
            same = dict(value=out["queue"]["value"], ms_per_run=out["queue"]["ms_per_run"], runs_ms=out["queue"]["runs_ms"], vs_resident=out["queue"]["vs_resident"],
                        note="the leg run in THIS process after the resident leg: the second family of contexts of a process runs 3-6 % slower at this size, whichever it is")
            out["queue"].update(value=fresh_queue["units"] / (f_ms * 1e-3), ms_per_run=f_ms, runs_ms=[round(x, 1) for x in fresh_queue["runs_ms"]],
                                batches=int(fresh_queue["batches"]), vs_resident=(fresh_queue["units"] / (f_ms * 1e-3)) / value,
                                process="a process of its own, started before this one touched the GPU (a caller of the queue holds no other contexts)",
                                same_process=same)
        q.close()
        ctx = capi.Context(local_rank)
        ctx.set_phase_groups(args.phase_groups)
        dchunks = []

    # ---- the shapes of BASELINE.json configs[2] and configs[4] on one GPU (same call, other chunks) ---------------
    if not single_process_multi:
        ctx.trim()
    if args.shape_runs > 0 and not single_process_multi:
        def shape_leg(name, what, make, n):
            with ThreadPoolExecutor(max_workers=n_threads) as ex:
                cs = list(ex.map(make, range(n)))
            for c in cs:
                capi.read_records(c)
            dcs = [capi.DeviceChunk.from_chunk(ctx, c) for c in cs]
            u = float(sum(c.units for c in cs))
            capi.phase_reads_many(ctx, dcs, cs, params, convert=False)
            barrier()
            t1, c1 = time.perf_counter(), time.process_time()
            for _ in range(args.shape_runs):
                _, sst = capi.phase_reads_many(ctx, dcs, cs, params, convert=False)
            barrier()
            el, cpu = time.perf_counter() - t1, time.process_time() - c1
            el, u_all = sharding.reduce_elapsed_and_units(dist, el, u, device=reduce_dev)
            # latency: ONE chunk and EIGHT chunks per call (what a `margin phase -t N` user with few chunks in hand sees)
            lat = {}
            for m in (1, 8):
                capi.phase_reads_many(ctx, dcs[:m], cs[:m], params, convert=False)
                t2 = time.perf_counter()
                for _ in range(5):
                    capi.phase_reads_many(ctx, dcs[:m], cs[:m], params, convert=False)
                lat[f"{m}_chunk{'s' if m > 1 else ''}"] = dict(ms_per_call=1e3 * (time.perf_counter() - t2) / 5,
                                                              value=float(sum(c.units for c in cs[:m])) * 5 / (time.perf_counter() - t2))
            for d_ in dcs:
                d_.close()
            # the path's roofline fraction for this shape: algorithmic bytes of every sweep of the call (24 B per cell, 32 B per merge
            # cell, 8 B per column; the profile bytes, a per cent of it, left out) over the wall time of the call
            alg_b = 24.0 * float(sst.cells) + 32.0 * float(sst.merge_cells) + 8.0 * float(sst.columns)
            ms_call = 1e3 * el / args.shape_runs
            out.setdefault("shapes", {})[name] = dict(what=what, chunks_per_gpu=n, value=u_all * args.shape_runs / el, unit="het-site-reads/s",
                                                     ms_per_call=ms_call, runs=args.shape_runs, resident=int(sst.resident),
                                                     fallback_chunks=int(sst.fallback_chunks), host_cpu_s_per_call=cpu / args.shape_runs,
                                                     path=dict(algorithmic_bytes=alg_b, achieved=alg_b / (ms_call * 1e-3) / 1e9, unit="GB/s",
                                                               frac=alg_b / (ms_call * 1e-3) / 1e9 / HBM_PEAK_GBS),
                                                     insitu=insitu_rooflines(sst), latency=lat)
        shape_leg("configs[2]", "chr20-like: 640 chunks of ~130 het sites (100 kb + margins), 30x ONT reads, one mrp_phase_reads_many call",
                  lambda s_: synth.make_ont_chunk(seed=50_000 + 1000 * rank + s_, region_bp=130 * 500, n_sites=130, coverage=args.coverage), 640)
        shape_leg("configs[4]", "HiFi-like: 48 chunks of 2 000 sites, 35x reads N(18 kb, 3 kb), 1 % allele error, 2-4 alleles per site "
                                "(shipped ONT haplotag parameters; phase_vcf mode differs in I/O only)",
                  lambda s_: synth.make_ont_chunk(seed=60_000 + 1000 * rank + s_, region_bp=args.sites * 500, n_sites=args.sites, coverage=35.0, median_len=18_000.0,
                                                  allele_error=0.01, allele_choices=(2, 3, 4), allele_probs=(0.85, 0.1, 0.05), length_model="normal",
                                                  normal_sd=3000.0), 48)

    # ---- kernel evidence: all sweeps of the same chunks replayed as one batch (rank 0) ---------------------------
    if not args.no_roofline and rank == 0 and not single_process_multi:
        rctx = capi.Context(local_rank)
        big = capi.Batch(rctx)
        tls_ctx = {}
        keep = []

        def record_one(i):
            import threading
            me = threading.get_ident()
            if me not in tls_ctx:
                tls_ctx[me] = capi.Context(local_rank)
            dch = capi.DeviceChunk.from_chunk(tls_ctx[me], chunks[i])
            keep.append(dch)
            return capi.phase_reads(tls_ctx[me], dch, chunks[i], params, record=big)["n_sweeps"]

        n_rec = min(args.roofline_chunks, args.chunks, len(chunks))
        t0 = time.time()
        with ThreadPoolExecutor(max_workers=n_threads) as ex:
            sweeps = sum(ex.map(record_one, range(n_rec)))
        t_build = time.time() - t0
        big.upload()
        for _ in range(3):
            big.launch()
        rctx.synchronize()
        big.stats()
        done, sw, em, pl = 0, 0.0, 0.0, 0.0
        t0 = time.perf_counter()
        while done < args.roofline_steps:  # the library keeps the events of a batch's 32 most recent launches
            k = min(32, args.roofline_steps - done)
            for _ in range(k):
                big.launch()
            s = big.stats()
            assert s.launches_averaged == k
            sw += s.avg_sweep_ms * k; em += s.avg_emission_ms * k; pl += s.avg_planes_ms * k
            done += k
        r_el = time.perf_counter() - t0
        s = big.stats()
        sweep_avg, emis_avg, planes_avg = sw / done, em / done, pl / done
        C, M, K = float(s.n_cells), float(s.n_merge_cells), float(s.n_columns)
        # B = sum_k 24*C_k + 32*M_k + D_k*Al_k + 8 per sweep.  The recursion kernel is credited with the
        # bytes it is responsible for (f and b 16*C, merge cells 32*M, column totals 8*K), the emission kernel with 8*C.
        alg = float(s.algorithmic_bytes)
        alg_sweep = 16.0 * C + 32.0 * M + 8.0 * K
        achieved = alg_sweep / (sweep_avg * 1e-3) / 1e9
        # HBM bytes of one launch from the PMC counters: measured by tools/collect_profiles_r03.sh on this workload and stamped
        # with the hash of the kernel source it was measured on -- a stale file is ignored rather than quoted
        traffic, traffic_src = None, None
        import hashlib
        k_sha = hashlib.sha256(open(os.path.join(ROOT, "margin_amd", "csrc", "mrp_kernels.hip"), "rb").read()).hexdigest()
        for tpath in (os.path.join(ROOT, "profiles", "r04", "traffic.json"),):
            if traffic is None and os.path.exists(tpath):
                tj = json.load(open(tpath))
                if int(tj.get("chunks", -1)) == n_rec and tj.get("kernel_source_sha256") == k_sha:
                    traffic, traffic_src = float(tj["sweep_kernel_hbm_bytes_per_launch"]), tj.get("source")
        replay_ms = 1e3 * r_el / done
        rec_units = float(sum(c.units for c in chunks[:n_rec]))
        out["roofline"] = dict(bound="hbm", kernel="mrp_sweep_i32_kernel", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                               # the same launch priced on the bytes that really crossed the HBM interface (the 32 B per merge
                               # cell of the algorithmic credit live in LDS): what the kernel sustains
                               frac_of_measured_traffic=(traffic / (sweep_avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                               frac_note="frac = ALGORITHMIC bytes (SURVEY.md 8d: 16 B per cell, 32 B per merge cell, 8 B per column) / kernel time / peak. "
                                         "traffic = HBM bytes of the same launch from the PMC counters: below the algorithmic credit when the merge cells' "
                                         "32 B stay in LDS, as they do (frac_of_measured_traffic prices those bytes instead)",
                               algorithmic_bytes_per_launch=alg_sweep, kernel_ms=sweep_avg,
                               what=f"all {sweeps} forward/backward sweeps of {n_rec} chunks (every merge level + final) recorded by the hashing path and "
                                    f"replayed as ONE dependency-free batch: the kernels' throughput, not a phasing rate",
                               moved_bytes_model=24.0 * C + 8.0 * M,
                               replay=dict(ms_per_launch=replay_ms, units_per_s=rec_units / (replay_ms * 1e-3), planes_ms=planes_avg, emission_ms=emis_avg,
                                           sweep_ms=sweep_avg, algorithmic_bytes=alg, frac=alg / (replay_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           emission_kernel=dict(algorithmic_bytes=8.0 * C, achieved=8.0 * C / (emis_avg * 1e-3) / 1e9),
                                           host_record_s=t_build),
                               # the path's fraction: algorithmic bytes of every sweep of a step over the wall time of the REAL step
                               path=dict(algorithmic_bytes=alg * args.chunks / n_rec, ms_per_step=ms_per_step,
                                         achieved=alg * args.chunks / n_rec / (ms_per_step * 1e-3) / 1e9,
                                         frac=alg * args.chunks / n_rec / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS))
        big.close()
        for d_ in keep:
            d_.close()
    if dist is not None:
        dist.barrier()

    if args.sum_chunks > 0 and rank == 0 and not single_process_multi:
        # Sum mode (logAddP with maxNotSumTransitions false, hmm.c:15-20): the sweeps of a few chunks phased by the hashing path
        # with log-sum-exp transitions, replayed as one batch: mrp_sweep_lse_kernel (merge column in LDS, reproducible sums).
        sctx = capi.Context(local_rank)
        sparams = capi.Params.from_reference_names(dict(params_dict, maxNotSumTransitions=0))
        sbatch = capi.Batch(sctx)
        skeep = []
        for i in range(min(args.sum_chunks, len(chunks))):
            dch = capi.DeviceChunk.from_chunk(sctx, chunks[i])
            skeep.append(dch)
            capi.phase_reads(sctx, dch, chunks[i], sparams, record=sbatch)
        sbatch.upload()
        sbatch.launch()
        sctx.synchronize()
        sbatch.stats()
        for _ in range(5):
            sbatch.launch()
        sctx.synchronize()
        ss = sbatch.stats()
        out["sum_mode"] = dict(what="log-sum-exp sweeps (maxNotSumTransitions = false) of the same chunks' merge levels, one batch; fp64, merge "
                                    "column in LDS, integer-atomic (order-free) accumulation",
                               chunks=len(skeep), hmms=int(ss.n_hmms), hmms_lds_kernel=int(ss.n_hmms_lse), hmms_generic_kernel=int(ss.n_hmms_generic),
                               cells=int(ss.n_cells), sweep_ms=float(ss.avg_sweep_ms), cells_per_s=float(ss.n_cells) / (ss.avg_sweep_ms * 1e-3),
                               dtype="f64", launches=int(ss.launches_averaged))
        sbatch.close()
        for d_ in skeep:
            d_.close()

    if args.align_chunks > 0 and not single_process_multi:
        # Alignment leg: the banded pair-HMM forward probability of every read substring against every
        # allele of every site (bubbleGraph.c:1421-1464), the numbers that become the profile bytes the sweeps read.
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
        # CPU baseline: the oracle (C restatement of the reference's linked-list/hash implementation, -O3 -mpopcnt) phasing
        # one chunk per thread -- the reference's own parallel axis (phase.c:276) -- on a bounded sample of the same chunks,
        # end to end like the GPU value (tiling paths, every merge level with prune, final sweep, trace back, fragments).
        from oracle import orc
        n_thr = max(1, min(16, os.cpu_count() or 1, len(chunks)))

        def cpu_one(c):
            oc = orc.OracleChunk(c)
            t1 = time.perf_counter()
            r = oc.phase(params_dict)
            dt = time.perf_counter() - t1
            oc.close()
            return dt, r["fb_seconds"], r["fb_calls"]

        t_cpu = time.perf_counter()
        with ThreadPoolExecutor(max_workers=n_thr) as ex:
            res = list(ex.map(cpu_one, chunks[:n_thr]))
        t_cpu = time.perf_counter() - t_cpu
        sample_units = sum(c.units for c in chunks[:n_thr])
        out["cpu_baseline"] = dict(value=sample_units / t_cpu, unit="het-site-reads/s", cores=n_thr, kind="port",
                                   sample=f"{n_thr} of {len(chunks)} chunks, one per thread, phased end to end by the oracle: {t_cpu:.1f} s wall "
                                          f"({sum(r[2] for r in res)} sweeps; {max(r[1] for r in res):.2f} s of the slowest thread inside forward/backward)",
                                   per_core=chunks[0].units / res[0][0])
    if rank == 0:
        print(json.dumps(out))
    if single_process_multi:
        queue.close()
    else:
        for d_ in dchunks:
            d_.close()
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
