#!/usr/bin/env python3
"""Benchmark of the stRPHmm forward/backward hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic 1 Mb chunks, 2 000 biallelic het sites, 30x ONT-like
reads, shipped ONT haplotag parameters (max-plus mode, 100 partitions per column).

One *step* = phasing every chunk of the GPU once, end to end, through the product's C-ABI (mrp_phase_reads_many):
tiling paths, every merge level (cross product -> forward/backward -> prune, resident in HBM), the final sweep with the
ancestor model, trace back, genome fragments and the read bipartition -- everything bubbleGraph_phaseBubbleGraph does from
profile sequences on.  Inputs (profile bytes, site tables) are resident in HBM when the timed region starts.
`value` = het-sites x reads of all chunks x steps / wall time: the rate at which chunks are actually phased.

Beside it:
  roofline      THE PATH: algorithmic bytes of every sweep of the timed step (24 B per cell, 32 B per merge cell, 8 B
                per column, counted by the engine while the step runs) over the wall time of the step, against the HBM peak;
                `traffic` = HBM bytes that really cross the interface per step, from the committed rocprofv3 PMC passes over every
                kernel of a 96-chunk batch in situ (profiles/r05/path_traffic.json, stamped with the kernel sources' hash);
                `insitu` = the same arithmetic per kernel family with the family's summed device time (HIP events of the step);
                `replay` = kernel evidence only: the recursion kernel over ALL sweeps of 96 chunks as one dependency-free batch.
  parity        the results of the LAST timed step for 16 of its chunks against the oracle's results for the same chunks (the
                cpu_baseline leg computes them anyway); the shape legs sample 8 chunks each the same way
  queue         the same chunks through the host work queue from HOST memory (upload included, PCIe-inclusive rate)
  shapes        the chunk shapes of BASELINE.json configs[2] (640 x ~130 sites), configs[4] (HiFi, 2-4 alleles) through the same call, and
                configs[3]/8: one GPU's share of a whole genome (3 900 chunks of ~130 sites) from HOST memory through the work queue
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
                    help="concurrent batches inside mrp_phase_reads_many (0: the library's choice, one per 12 chunks up to 4, eight from 192 chunks on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the kernel replay leg (it needs ~20 s of host work to record the sweeps)")
    ap.add_argument("--roofline-steps", type=int, default=20)
    ap.add_argument("--roofline-chunks", type=int, default=96, help="chunks whose sweeps the kernel replay leg records (host work: ~0.2 s per chunk)")
    ap.add_argument("--queue-runs", type=int, default=3, help="runs of the host-memory work queue leg (0: skip)")
    ap.add_argument("--queue-batch", type=int, default=0, help="chunks per batch of the queue leg (0: the library's choice: one call per device up to 1 280 chunks of this kind, batches of at most 192 beyond)")
    ap.add_argument("--shape-runs", type=int, default=3, help="runs of the configs[2] / configs[4] shape legs (0: skip)")
    ap.add_argument("--align-chunks", type=int, default=4, help="chunks whose read x allele pairs the alignment leg scores (0: skip)")
    ap.add_argument("--align-runs", type=int, default=3)
    ap.add_argument("--sum-chunks", type=int, default=16, help="chunks whose sweeps the log-sum-exp leg replays (0: skip)")
    ap.add_argument("--genome-chunks", type=int, default=3900, help="chunks of the configs[3]/8 leg: one GPU's eighth of a 31 000-chunk genome (0: skip)")
    ap.add_argument("--parity-chunks", type=int, default=16, help="chunks of the timed step whose results are compared with the oracle's (needs the cpu_baseline leg)")
    ap.add_argument("--queue-child", action="store_true", help="(internal) run the work queue leg alone and print its runs as one JSON line")
    ap.add_argument("--queue-device", type=int, default=0, help="(internal) device of the queue child")
    return ap.parse_args()


def queue_child(args):
    """The work queue leg in a process of its own (started by the parent before IT touches the GPU): a caller of the queue holds no
    other family of contexts, and the queue's workers bring contexts, streams, a host pool and allocator caches of their own --
    whichever of the two families of a process comes second runs 3-6 % slower at 1 152 chunks (measured both ways round)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    from margin_amd import capi, sharding, synth
    params = capi.Params.from_reference_names(synth.shipped_phase_params())
    cpu_share = max(1, os.cpu_count() or 8)
    capi.load().mrp_set_host_threads(max(1, min(int(os.environ.get("MRP_BENCH_HOST_THREADS", "32")), cpu_share)))
    n_threads = args.threads or min(16, cpu_share, args.chunks)
    seeds = sharding.chunk_seeds(0, args.chunks)
    with ThreadPoolExecutor(max_workers=n_threads) as ex:
        chunks = list(ex.map(lambda sd: synth.make_ont_chunk(seed=sd, region_bp=args.sites * 500, n_sites=args.sites, coverage=args.coverage), seeds))
    for c in chunks:
        capi.read_records(c)
    descs = capi.chunk_descs(chunks)
    q = capi.Queue([args.queue_device])
    for _ in range(2):
        q.phase(chunks, params, chunks_per_batch=args.queue_batch, descs=descs, convert=False)
    runs = []
    for _ in range(args.queue_runs):
        t0 = time.perf_counter()
        _, qst = q.phase(chunks, params, chunks_per_batch=args.queue_batch, descs=descs, convert=False)
        runs.append(1e3 * (time.perf_counter() - t0))
    q.close()
    print(json.dumps(dict(queue_child=True, runs_ms=runs, batches=int(qst.batches), units=float(sum(c.units for c in chunks)))))


def run_queue_child(args, device):
    """parent side -> (result dict or None, how the leg was run: "child" or "child-failed: <reason>")"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--queue-child", "--chunks", str(args.chunks), "--sites", str(args.sites),
           "--coverage", str(args.coverage), "--queue-runs", str(args.queue_runs), "--queue-batch", str(args.queue_batch), "--threads", str(args.threads),
           "--queue-device", str(device)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        return None, "child-failed: no result within 300 s"
    except OSError as e:
        return None, f"child-failed: {type(e).__name__}: {e}"
    if res.returncode != 0:
        return None, f"child-failed: exit code {res.returncode}: {res.stderr.strip().splitlines()[-1][:200] if res.stderr.strip() else 'no stderr'}"
    for line in reversed(res.stdout.splitlines()):
        if line.startswith("{") and "queue_child" in line:
            return json.loads(line), "child"
    return None, "child-failed: no result line on stdout"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


PHASE_KEYS = ("hap1", "hap2", "genotype", "ancestor", "support1", "support2", "genotype_probs", "hap_probs1", "hap_probs2")


def same_phasing(a, b):
    """haplotype strings, genotypes, per-site supports and probabilities, and the read bipartition in order: bit for bit"""
    import numpy as np
    return (a["ref_start"], a["length"]) == (b["ref_start"], b["length"]) and all((np.asarray(a[k]) == np.asarray(b[k])).all() for k in PHASE_KEYS) \
        and a["reads1"] == b["reads1"] and a["reads2"] == b["reads2"]


def oracle_phase_sample(chunks, params_dict, n_threads):
    """the oracle (test infrastructure: the checker and the CPU baseline, never the product) over these chunks, one per thread ->
    (results, wall seconds, [(seconds, seconds inside forward/backward, sweeps) per chunk])"""
    from oracle import orc

    def one(c):
        oc = orc.OracleChunk(c)
        t1 = time.perf_counter()
        r = oc.phase(params_dict)
        dt = time.perf_counter() - t1
        oc.close()
        return r, (dt, r["fb_seconds"], r["fb_calls"])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, n_threads)) as ex:
        res = list(ex.map(one, chunks))
    return [r[0] for r in res], time.perf_counter() - t0, [r[1] for r in res]


def path_traffic(n_chunks_per_step, workload_sites):
    """HBM bytes that really cross the interface per step: sum over EVERY kernel of one 96-chunk batch in situ of (2 x FETCH_SIZE +
    WRITE_SIZE) x 1024 (separate --pmc passes; gfx950 tallies a wide coalesced read at one half, MI355X_MICROARCH.md), scaled by the
    step's chunks.  Read from the committed summary of the round's profile run, which is stamped with the hash of the kernel sources
    it was measured on: a stale file is ignored rather than quoted.  -> (bytes per step or None, source)"""
    import hashlib
    tpath = os.path.join(ROOT, "profiles", "r05", "path_traffic.json")
    if not os.path.exists(tpath) or workload_sites != 2000:
        return None, None
    tj = json.load(open(tpath))
    h = hashlib.sha256()
    for f in ("mrp_kernels.hip", "mrp_engine_kernels.hip"):
        h.update(open(os.path.join(ROOT, "margin_amd", "csrc", f), "rb").read())
    if tj.get("kernel_sources_sha256") != h.hexdigest():
        return None, "profiles/r05/path_traffic.json is older than the kernel sources: not quoted"
    return float(tj["hbm_bytes_per_batch"]) * n_chunks_per_step / float(tj["chunks"]), tj.get("source")


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
    fresh_queue, queue_process = None, "in-process"
    if world == 1 and n_gpus == 1 and args.queue_runs > 0 and args.steps > 0 and "MRP_BENCH_DEVICES" not in os.environ:
        fresh_queue, queue_process = run_queue_child(args, local_rank)
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
    # the library's worker pool: this rank's share of the node's cores, at most 32 per device (a work queue gives every
    # device its own pool of this size).  Round 5: 32 instead of 16 -- the GPU hosts of this pool have 256 hardware threads for eight
    # devices, and the host-bound head of a call (the first merge levels' tiling paths) shrinks with the pool: 157 -> 153 ms per step
    # at 32 threads, 152 at 48 (tools/step_probe.py); the pool's threads sleep outside their loops
    host_threads = max(1, min(int(os.environ.get("MRP_BENCH_HOST_THREADS", "32")), cpu_share // (n_gpus if single_process_multi else 1)))
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
    # parity stamp: the results of the LAST timed step for the first chunks stay unconverted until the clock has stopped
    n_parity = min(args.parity_chunks, len(chunks)) if (rank == 0 and not args.no_cpu_baseline and n_gpus == 1) else 0
    if single_process_multi:
        queue = capi.Queue(queue_devices)
        descs = capi.chunk_descs(chunks)
        step = lambda defer=(): queue.phase(chunks, params, chunks_per_batch=args.queue_batch, descs=descs, convert=False, defer=defer)
    else:
        ctx = capi.Context(local_rank)
        ctx.set_phase_groups(args.phase_groups)
        dchunks = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
        many_args = capi.phase_many_args(dchunks, chunks)  # (the call's argument arrays: built once, not per step)
        step = lambda defer=(): capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=many_args, defer=defer)
    for _ in range(args.warmup):
        step()
    barrier()
    t0, c0 = time.perf_counter(), time.process_time()
    st, step_results, step_ms = None, None, []
    for k_ in range(args.steps):
        t_s = time.perf_counter()
        step_results, st = step(range(n_parity) if k_ + 1 == args.steps else ())
        step_ms.append(1e3 * (time.perf_counter() - t_s))
    barrier()
    elapsed, host_cpu = time.perf_counter() - t0, time.process_time() - c0
    elapsed, units_all = sharding.reduce_elapsed_and_units(dist, elapsed, units, device=reduce_dev)
    gpu_sample = [step_results[i].get() for i in range(n_parity)] if step_results is not None else []
    step_results = None
    # (--steps 0: the counter passes of tools/collect_profiles_r05.sh, which want the replay leg's launches alone in the trace)
    value = units_all * args.steps / elapsed if args.steps > 0 else 0.0
    ms_per_step = 1e3 * elapsed / args.steps if args.steps > 0 else float("nan")
    median_step_ms = sorted(step_ms)[len(step_ms) // 2] if step_ms else float("nan")

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
        out["step_detail"]["median_step_ms"] = median_step_ms
        # ---- roofline: THE PATH.  Algorithmic bytes of every sweep of the step (B = sum 24 C_k + 32 M_k + 8 per column; the
        # profile bytes D_k Al_k, under one per cent, are left out), counted by the engine on the hmms it really built, over the wall
        # time of the step.  (The engine's counters are those of the last step; every step phases the same chunks.)
        if args.steps > 0:
            alg_step = 24.0 * float(st.cells) + 32.0 * float(st.merge_cells) + 8.0 * float(st.columns)
            ach = alg_step / (ms_per_step * 1e-3) / 1e9
            traffic, traffic_src = path_traffic(args.chunks, args.sites)
            out["roofline"] = dict(bound="hbm", kernel="the whole step: every kernel of mrp_phase_reads_many (dominant families under `insitu`)",
                                   achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                                   frac_of_measured_traffic=(traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                                   algorithmic_bytes_per_step=alg_step, ms_per_step=ms_per_step,
                                   frac_note="frac = ALGORITHMIC bytes of the step (24 B per cell, 32 B per merge cell, 8 B per column, SURVEY.md 8d) / wall time of "
                                             "the step / HBM peak: the path's figure, end to end.  traffic = HBM bytes that really cross the interface per step "
                                             "(PMC, every kernel in situ; the merge cells' 32 B live in LDS and the level arrays hold one entry per complement pair, "
                                             "so it is below the algorithmic credit).  insitu: per kernel family, its algorithmic bytes over its summed device time in "
                                             "the step.  replay: kernel evidence only (a dependency-free launch of the recursion kernel), not the path",
                                   insitu=insitu_rooflines(st))
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
        # both legs by the same statistic: the MEDIAN run of the queue over the MEDIAN step of the resident leg (vs_resident); the ratio to the
        # headline value (a mean over the steps, as the contract asks) is listed beside it
        res_rate = units_all / (median_step_ms * 1e-3) if world == 1 and args.steps > 0 else value
        out["queue"] = dict(what="mrp_queue_phase_chunks: the same chunks from HOST memory (PCIe-inclusive): sorted by estimated cost, pulled in "
                                 "batches by one worker per device, the next batch's site tables and profile bytes uploaded on a second stream "
                                 "while the current batch is phased",
                            value=q_units / q_el, unit="het-site-reads/s", ms_per_run=1e3 * q_el, runs_ms=[round(x, 1) for x in q_ms],
                            batches=int(qst.batches), runs=args.queue_runs, vs_resident=(q_units / q_el) / res_rate, vs_value=(q_units / q_el) / value,
                            statistic="median run / median step of the resident leg", process=queue_process)
        if fresh_queue is not None and fresh_queue.get("runs_ms"):
            # the same leg from a process of its own (run_queue_child): that is what a caller of the queue sees; the in-process runs above
            # come second in a process whose first family of contexts -- closed by now -- was the resident leg's
            f_ms = sorted(fresh_queue["runs_ms"])[len(fresh_queue["runs_ms"]) // 2]
            same = dict(value=out["queue"]["value"], ms_per_run=out["queue"]["ms_per_run"], runs_ms=out["queue"]["runs_ms"], vs_resident=out["queue"]["vs_resident"],
                        note="the leg run in THIS process after the resident leg: the second family of contexts of a process runs 3-6 % slower at this size, whichever it is")
            f_rate = fresh_queue["units"] / (f_ms * 1e-3)
            out["queue"].update(value=f_rate, ms_per_run=f_ms, runs_ms=[round(x, 1) for x in fresh_queue["runs_ms"]],
                                batches=int(fresh_queue["batches"]), vs_resident=f_rate / res_rate, vs_value=f_rate / value,
                                process="child: a process of its own on the same device, started before this one touched the GPU (a caller of the queue holds no other "
                                        "contexts); median of its runs after two warm-up calls",
                                same_process=same)
        q.close()
        ctx = capi.Context(local_rank)
        ctx.set_phase_groups(args.phase_groups)
        dchunks = []

    # ---- the shapes of BASELINE.json configs[2] and configs[4] on one GPU (same call, other chunks) ---------------
    if not single_process_multi:
        ctx.trim()
    if args.shape_runs > 0 and not single_process_multi:
        do_parity = rank == 0 and not args.no_cpu_baseline and n_gpus == 1

        def parity_of(cs, got, idx):
            """the oracle over the sampled chunks against the device results of the leg's last call"""
            ref, _, _ = oracle_phase_sample([cs[i] for i in idx], params_dict, min(16, os.cpu_count() or 1))
            ok = sum(1 for r, i in zip(ref, idx) if same_phasing(got[i].get(), r))
            return dict(identical=f"{ok}/{len(idx)}", chunks=[int(i) for i in idx],
                        against="oracle/rphmm_oracle.c phasing the same chunks end to end (haplotype strings, genotypes, supports, probabilities, read bipartition)")

        def shape_leg(name, what, make, n, sample=8):
            with ThreadPoolExecutor(max_workers=n_threads) as ex:
                cs = list(ex.map(make, range(n)))
            for c in cs:
                capi.read_records(c)
            dcs = [capi.DeviceChunk.from_chunk(ctx, c) for c in cs]
            u = float(sum(c.units for c in cs))
            sargs = capi.phase_many_args(dcs, cs)  # (the ctypes argument arrays, built once as for the headline leg: not the library's time)
            for _ in range(2):
                capi.phase_reads_many(ctx, dcs, cs, params, convert=False, prepared=sargs)
            idx = sorted({(j * max(1, n // sample)) % n for j in range(sample)}) if do_parity else []
            barrier()
            t1, c1 = time.perf_counter(), time.process_time()
            for r_ in range(args.shape_runs):
                got, sst = capi.phase_reads_many(ctx, dcs, cs, params, convert=False, prepared=sargs, defer=idx if r_ + 1 == args.shape_runs else ())
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
            if idx:
                out["shapes"][name]["parity"] = parity_of(cs, got, idx)
            return cs, alg_b

        small = lambda s_: synth.make_ont_chunk(seed=50_000 + 1000 * rank + s_, region_bp=130 * 500, n_sites=130, coverage=args.coverage)
        shape_leg("configs[2]", "chr20-like: 640 chunks of ~130 het sites (100 kb + margins), 30x ONT reads, one mrp_phase_reads_many call", small, 640)
        shape_leg("configs[4]", "HiFi-like: 48 chunks of 2 000 sites, 35x reads N(18 kb, 3 kb), 1 % allele error, 2-4 alleles per site "
                                "(shipped ONT haplotag parameters; phase_vcf mode differs in I/O only)",
                  lambda s_: synth.make_ont_chunk(seed=60_000 + 1000 * rank + s_, region_bp=args.sites * 500, n_sites=args.sites, coverage=35.0, median_len=18_000.0,
                                                  allele_error=0.01, allele_choices=(2, 3, 4), allele_probs=(0.85, 0.1, 0.05), length_model="normal",
                                                  normal_sd=3000.0), 48)
        if args.genome_chunks > 0:
            # configs[3]/8: one GPU's share of a whole genome -- BASELINE.json configs[3] is ~31 000 chunks of 100 kb (htsIntegration.c:151-179) over
            # eight GPUs -- from HOST memory through the work queue on this device: what a `margin phase` run over a genome asks of one GPU.
            # The bytes of the path come from one resident call over the same chunks (the same hmms; the queue's statistics do not carry them).
            n_g = args.genome_chunks
            with ThreadPoolExecutor(max_workers=n_threads) as ex:
                gcs = list(ex.map(small, range(n_g)))
            for c in gcs:
                capi.read_records(c)
            gu = float(sum(c.units for c in gcs))
            gd = [capi.DeviceChunk.from_chunk(ctx, c) for c in gcs]
            gargs = capi.phase_many_args(gd, gcs)
            for _ in range(2):
                capi.phase_reads_many(ctx, gd, gcs, params, convert=False, prepared=gargs)
            t1 = time.perf_counter()
            _, gst = capi.phase_reads_many(ctx, gd, gcs, params, convert=False, prepared=gargs)
            g_res_ms = 1e3 * (time.perf_counter() - t1)
            for d_ in gd:
                d_.close()
            ctx.trim()
            g_alg = 24.0 * float(gst.cells) + 32.0 * float(gst.merge_cells) + 8.0 * float(gst.columns)
            gq = capi.Queue([local_rank])
            gdescs = capi.chunk_descs(gcs)
            gq.phase(gcs, params, chunks_per_batch=0, descs=gdescs, convert=False)
            gidx = sorted({(j * max(1, n_g // 16)) % n_g for j in range(16)}) if do_parity else []
            barrier()
            g_ms, g_cpu0 = [], time.process_time()
            for r_ in range(max(1, args.shape_runs)):
                t1 = time.perf_counter()
                ggot, gqst = gq.phase(gcs, params, chunks_per_batch=0, descs=gdescs, convert=False, defer=gidx if r_ + 1 == max(1, args.shape_runs) else ())
                barrier()
                g_ms.append(1e3 * (time.perf_counter() - t1))
            g_cpu = (time.process_time() - g_cpu0) / len(g_ms)
            g_el = sorted(g_ms)[len(g_ms) // 2] * 1e-3
            g_el, gu_all = sharding.reduce_elapsed_and_units(dist, g_el, gu, device=reduce_dev)
            out["shapes"]["configs[3]/8"] = dict(
                what=f"one GPU's eighth of a whole genome: {n_g} chunks of ~130 het sites, 30x ONT reads, from HOST memory through mrp_queue_phase_chunks on one device "
                     "(upload inside the timed region; batches cut by units, four lanes)",
                chunks_per_gpu=n_g, value=gu_all / g_el, unit="het-site-reads/s", ms_per_run=1e3 * g_el, runs_ms=[round(x, 1) for x in g_ms], batches=int(gqst.batches),
                fallback_chunks=int(gqst.fallback_chunks), host_cpu_s_per_run=g_cpu,
                path=dict(algorithmic_bytes=g_alg, achieved=g_alg / g_el / 1e9, unit="GB/s", frac=g_alg / g_el / 1e9 / HBM_PEAK_GBS),
                resident_call=dict(ms_per_call=g_res_ms, value=gu / (g_res_ms * 1e-3), note="the same chunks resident in HBM, ONE mrp_phase_reads_many call"))
            if gidx:
                out["shapes"]["configs[3]/8"]["parity"] = parity_of(gcs, ggot, gidx)
            gq.close()

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
        # HBM bytes of one launch from the PMC counters: measured by tools/collect_profiles_r05.sh on this workload and stamped
        # with the hash of the kernel source it was measured on -- a stale file is ignored rather than quoted
        traffic, traffic_src = None, None
        import hashlib
        k_sha = hashlib.sha256(open(os.path.join(ROOT, "margin_amd", "csrc", "mrp_kernels.hip"), "rb").read()).hexdigest()
        for tpath in (os.path.join(ROOT, "profiles", "r05", "traffic.json"), os.path.join(ROOT, "profiles", "r04", "traffic.json")):
            if traffic is None and os.path.exists(tpath):
                tj = json.load(open(tpath))
                if int(tj.get("chunks", -1)) == n_rec and tj.get("kernel_source_sha256") == k_sha:
                    traffic, traffic_src = float(tj["sweep_kernel_hbm_bytes_per_launch"]), tj.get("source")
        replay_ms = 1e3 * r_el / done
        rec_units = float(sum(c.units for c in chunks[:n_rec]))
        replay = dict(kernel="mrp_sweep_i32_kernel", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                      traffic=traffic, traffic_source=traffic_src,
                      # the same launch priced on the bytes that really crossed the HBM interface (the 32 B per merge
                      # cell of the algorithmic credit live in LDS): what the kernel sustains
                      frac_of_measured_traffic=(traffic / (sweep_avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                      frac_note="KERNEL EVIDENCE, NOT THE PATH: ALGORITHMIC bytes (SURVEY.md 8d: 16 B per cell, 32 B per merge cell, 8 B per column) of ONE "
                                "dependency-free launch of the recursion kernel / its HIP-event time / peak.  traffic = HBM bytes of the same launch from the PMC counters",
                      algorithmic_bytes_per_launch=alg_sweep, kernel_ms=sweep_avg,
                      what=f"all {sweeps} forward/backward sweeps of {n_rec} chunks (every merge level + final) recorded by the hashing path and "
                           f"replayed as ONE dependency-free batch on per-cell arrays: the kernel's throughput, not a phasing rate",
                      moved_bytes_model=24.0 * C + 8.0 * M, ms_per_launch=replay_ms, units_per_s=rec_units / (replay_ms * 1e-3), planes_ms=planes_avg,
                      emission_ms=emis_avg, sweep_ms=sweep_avg, algorithmic_bytes=alg, whole_launch_frac=alg / (replay_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      emission_kernel=dict(algorithmic_bytes=8.0 * C, achieved=8.0 * C / (emis_avg * 1e-3) / 1e9), host_record_s=t_build)
        if "roofline" in out:
            out["roofline"]["replay"] = replay
        else:  # (--steps 0: the counter passes of the profile script run the replay alone)
            out["roofline"] = dict(bound="hbm", kernel="mrp_sweep_i32_kernel (replay only: no step was timed)", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                                   frac=achieved / HBM_PEAK_GBS, traffic=traffic, replay=replay)
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
        n_thr = max(1, min(16, os.cpu_count() or 1, len(chunks)))
        n_cpu = max(n_thr, n_parity)
        ref_res, t_cpu, res = oracle_phase_sample(chunks[:n_cpu], params_dict, n_thr)
        sample_units = sum(c.units for c in chunks[:n_cpu])
        out["cpu_baseline"] = dict(value=sample_units / t_cpu, unit="het-site-reads/s", cores=n_thr, kind="port",
                                   sample=f"{n_cpu} of {len(chunks)} chunks, one per thread, phased end to end by the oracle: {t_cpu:.1f} s wall "
                                          f"({sum(r[2] for r in res)} sweeps; {max(r[1] for r in res):.2f} s of the slowest thread inside forward/backward)",
                                   per_core=chunks[0].units / res[0][0], cpu_model=cpu_model(), host_cpus=os.cpu_count(),
                                   usable_cpus=len(os.sched_getaffinity(0)))
        # parity stamp of the TIMED STEP: its results for the same chunks (taken from the last timed step, converted after the clock stopped)
        if gpu_sample:
            ok = sum(1 for g_, r_ in zip(gpu_sample, ref_res) if same_phasing(g_, r_))
            out["parity"] = dict(identical=f"{ok}/{len(gpu_sample)}", chunks=list(range(len(gpu_sample))),
                                 what=f"results of the last timed step ({args.chunks} chunks in one mrp_phase_reads_many call) for its first {len(gpu_sample)} chunks against "
                                      "oracle/rphmm_oracle.c phasing the same chunks end to end: haplotype strings, genotypes, supports, probabilities and the read "
                                      "bipartition (HP tags) in order, bit for bit",
                                 shapes={k: v["parity"]["identical"] for k, v in out.get("shapes", {}).items() if "parity" in v},
                                 alignment=out.get("alignment", {}).get("cpu_baseline", {}).get("identical_to_gpu"))
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
