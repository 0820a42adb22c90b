"""GPU: the N > 1 pipeline (tiling, `svr_gather_tiles` over RCCL, un-tile) in a one-rank process group — RCCL refuses
two ranks on one device, so this is what one GPU can run; world-size 2-4 runs are covered on CPUs
(tests/test_distributed_cpu.py).  Each scenario runs in a process of its own (tests/collective_worker.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run(scenario):
    with socket.socket() as sock:                        # a port nobody holds right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(HERE, "collective_worker.py"), scenario],
                         cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_gather_on_one_stream_finish_on_another_never_reads_early():
    """`gather_async` leaves the transfers on the render's stream; `finish` on ANOTHER stream must wait for them before it
    un-tiles, and the caller may overwrite the region buffers right after `finish` (an event per slot orders both)."""
    rep = run("streams")
    assert rep["init_comm"] is True and rep["transport"].startswith("svr_gather_tiles")
    assert rep["frames_equal_single_gpu_render"] == [True] * 6, rep


@pytest.mark.parametrize("scenario", ["probe_fails", "no_rccl"])
def test_fallback_to_torch_distributed_gather_reassembles_the_same_frame(scenario):
    """`init_comm` checks its communicator with one probe gather; when the probe (or the communicator itself) fails on
    any rank, every rank falls back to `torch.distributed.gather` — and that path gives the same frame."""
    rep = run(scenario)
    assert rep["init_comm"] is False and rep["transport"] == "torch.distributed.gather"
    assert rep["frames_equal_single_gpu_render"] == [True] * 6, rep
