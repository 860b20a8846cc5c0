#!/usr/bin/env python3
"""Development probe: the headline step (one mrp_phase_reads_many call over N configs[1] chunks, inputs resident) under a list of
settings that can be changed at run time, chunks synthesised ONCE.  usage: step_probe.py [--chunks 1152] [--steps 6] SETTING ...
  SETTING = comma-separated KEY=VALUE: threads=<host pool threads>, groups=<concurrent batches>, or any environment variable the library
  reads per call (MRP_...).  Prints the median and all step times per setting."""
import argparse
import ctypes
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from margin_amd import capi, synth  # noqa: E402


def cpu_stat():
    """CPU bandwidth control of the container (cgroup v2): periods in which the process group was throttled, and for how long"""
    try:
        return {k: int(v) for k, v in (line.split() for line in open("/sys/fs/cgroup/cpu.stat"))}
    except (OSError, ValueError):
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=1152)
    ap.add_argument("--sites", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("settings", nargs="*", default=["threads=16"])
    args = ap.parse_args()
    params = capi.Params.from_reference_names(synth.shipped_phase_params())
    with ThreadPoolExecutor(max_workers=16) as ex:
        chunks = list(ex.map(lambda s: synth.make_ont_chunk(seed=s + 1, region_bp=args.sites * 500, n_sites=args.sites, coverage=30.0), range(args.chunks)))
    units = sum(c.units for c in chunks)
    for c in chunks:
        capi.read_records(c)
    capi.load().mrp_set_host_threads(16)
    ctx = capi.Context(0)
    dchunks = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
    prepared = capi.phase_many_args(dchunks, chunks)
    for _ in range(2):
        capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=prepared)
    all_cpus = os.sched_getaffinity(0)
    for setting in args.settings:
        saved = {}
        for kv in setting.split(","):
            k, v = kv.split("=", 1)
            if k == "threads":
                capi.load().mrp_set_host_threads(int(v))
            elif k == "groups":
                ctx.set_phase_groups(int(v))
            elif k == "cpuprofile":
                pass
            elif k == "cpus":  # CPU affinity of every thread of the process, e.g. cpus=0-15 or cpus=0-7+128-135 ("all": 0-4095)
                cpus = set()
                for part in ("0-4095" if v == "all" else v).split("+"):
                    lo, _, hi = part.partition("-")
                    cpus.update(range(int(lo), int(hi or lo) + 1))
                cpus &= all_cpus
                for tid in os.listdir("/proc/self/task"):
                    try:
                        os.sched_setaffinity(int(tid), cpus)
                    except OSError:
                        pass
            else:
                saved[k] = os.environ.get(k)
                os.environ[k] = v
        capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=prepared)
        ms, cpu0, thr0 = [], time.process_time(), cpu_stat()
        lib = capi.load()
        lib.mrp_pool_tag_cpu_ns.restype = ctypes.c_longlong
        tag0 = [lib.mrp_pool_tag_cpu_ns(t) for t in range(16)]
        for _ in range(args.steps):
            t0 = time.perf_counter()
            _, st = capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=prepared)
            ms.append(1e3 * (time.perf_counter() - t0))
        if "cpuprofile=1" in setting:  # CPUs in use over the course of one more step: process CPU time sampled every 2 ms
            import threading
            samples, stop = [], threading.Event()

            def sampler():
                while not stop.is_set():
                    samples.append((time.perf_counter(), time.process_time()))
                    time.sleep(0.002)
            th = threading.Thread(target=sampler)
            th.start()
            time.sleep(0.01)
            t0 = time.perf_counter()
            capi.phase_reads_many(ctx, dchunks, chunks, params, convert=False, prepared=prepared)
            t1 = time.perf_counter()
            time.sleep(0.01)
            stop.set()
            th.join()
            print(f"  CPUs in use per 10 ms of a {1e3 * (t1 - t0):.1f} ms step:", end="")
            b = 0
            while t0 + 0.01 * b < t1:
                lo, hi = t0 + 0.01 * b, t0 + 0.01 * (b + 1)
                a = [x for x in samples if x[0] <= lo]
                z = [x for x in samples if x[0] >= hi]
                if a and z:
                    print(f" {(z[0][1] - a[-1][1]) / (z[0][0] - a[-1][0]):.1f}", end="")
                b += 1
            print(flush=True)
        cpu = (time.process_time() - cpu0) / args.steps
        med = sorted(ms)[len(ms) // 2]
        thr1 = cpu_stat()
        names = {0: "other", 1: "prepare", 2: "finish", 3: "gather", 4: "setup", 5: "final_shadow", 6: "many_finish", 8: "stage"}
        tags = "  pool cpu ms/step: " + " ".join(f"{names.get(t, t)} {(lib.mrp_pool_tag_cpu_ns(t) - tag0[t]) / 1e6 / args.steps:.0f}" for t in range(16) if lib.mrp_pool_tag_cpu_ns(t) - tag0[t] > 0)
        thr = f"  cgroup: {thr1.get('nr_throttled', 0) - thr0.get('nr_throttled', 0)} throttled periods, {(thr1.get('throttled_usec', 0) - thr0.get('throttled_usec', 0)) / 1e3:.0f} ms" if thr1 else ""
        print(f"{setting:40s} median {med:7.1f} ms = {units / med / 1e3:.3e} units/s  host cpu {cpu:.2f} s  runs {[round(x, 1) for x in ms]}  "
              f"(pack {st.pack_ms:.0f} xe {st.cross_emit_ms:.0f} rec {st.recursion_ms:.0f} prune {st.prune_kernel_ms:.0f} compact {st.compact_ms:.0f}){thr}{tags}", flush=True)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        capi.load().mrp_set_host_threads(16)
        ctx.set_phase_groups(0)


if __name__ == "__main__":
    main()
