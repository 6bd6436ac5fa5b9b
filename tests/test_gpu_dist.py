"""-m gpu: the multi-rank path on real kernels.  `gpurun` boxes have ONE GPU, so both ranks share cuda:0 and the process group is
gloo (RCCL refuses two ranks on one device): everything of SURVEY.md 8e except the xGMI transport itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "4"
    return env


@pytest.mark.parametrize("name", ["Tiny", "EfficientConformerCTCLarge"])      # Large = BASELINE.json configs[2], the data-parallel headline
def test_sharded_encoder_two_ranks_on_one_gpu_equals_unsharded(name):
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), name],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert "DIST_GPU_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("name", ["Tiny", "EfficientConformerCTCLarge"])      # Large = BASELINE.json configs[2] (the driver-run test was Tiny only)
def test_sharded_encoder_one_rank_over_rccl(name):
    """The same worker on the REAL backend: backend "nccl" = RCCL, one rank (a `gpurun` box has one GPU and RCCL takes one rank per device).
    No xGMI transfer happens, but everything else of the N > 1 path runs on RCCL's own machinery: communicator creation on the MI355X, collectives
    enqueued from the row ranges' streams onto the process group's stream, `async_op` Work objects and their stream-side wait() in the pipelined
    protocol, allocator stream bookkeeping of the gathered chunks - with the bit-for-bit assertions of the two-rank gloo run."""
    env = _env()
    env["EFFCONF_TEST_BACKEND"] = "nccl"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), name],
                       capture_output=True, text=True, timeout=600, env=env)
    assert "DIST_GPU_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("gather", ["outputs", "labels"])
def test_bench_two_ranks_on_one_gpu(gather):
    """`python bench.py --gpus 2` end to end (launcher, per-range all-gather on the comm stream, head on the gathered chunks on the head
    stream, max-over-ranks timing), both ranks on cuda:0 over gloo."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device", "--batch", "16",
                        "--steps", "3", "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--gather", gather],
                       capture_output=True, text=True, timeout=600, env=_env())
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 32 and j["value"] > 0
    assert j["check"]["ok"], j["check"]                      # rank 0 verifies what it holds after the last step (gathered chunks / label ids)
