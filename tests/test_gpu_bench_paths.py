"""Both multi-GPU code paths of bench.py executed once on the ONE card a test box has: the driver's
launch -- one rank per GPU under torch.distributed.run, chunks sharded by rank, barrier + MAX-over-ranks timing -- rehearsed with
two ranks on device 0 over gloo, and the single-process path (--gpus N, the library's work queue over N devices) over the device
list [0, 0].  Children are fresh processes; nothing here replaces a process that has touched the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS_OFF = ["--no-cpu-baseline", "--no-roofline", "--queue-runs", "0", "--shape-runs", "0", "--align-chunks", "0", "--sum-chunks", "0"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_two_ranks_on_one_card_over_gloo():
    env = dict(os.environ, MRP_BENCH_BACKEND="gloo", MRP_BENCH_DEVICE="0", MRP_QUIET="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--chunks", "96"] + LEGS_OFF
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["chunks_per_gpu"] == 96 and "2 process(es), one per GPU" in d["config"]["parallelism"]
    assert d["step_detail"]["resident"] == 1 and d["step_detail"]["fallback_chunks"] == 0
    # whole-job aggregate: both ranks' units over the slower rank's time -- more than one rank's units over that time
    assert d["value"] * d["ms_per_step"] * 1e-3 > 1.5 * d["config"]["units_per_gpu"]


def test_bench_single_process_queue_over_two_workers_on_one_card():
    env = dict(os.environ, MRP_BENCH_DEVICES="0,0", MRP_QUIET="1")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--chunks", "48"] + LEGS_OFF
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "host work queue" in d["config"]["parallelism"]
    sd = d["step_detail"]
    assert sum(sd["chunks_per_device"]) == 96 and sd["fallback_chunks"] == 0 and sd["batches"] >= 2
    assert min(sd["chunks_per_device"]) > 0  # both workers took work (the first batch of every device is fixed)
