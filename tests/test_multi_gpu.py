"""GPU tests (-m gpu) of the tile-row sharded path with MORE THAN ONE RANK running the HIP kernels (north_star: frames shard by
screen-tile row with an RCCL gather of the image).

  * two ranks on ONE GPU, gloo backend (the exchange code stages device tensors through the host): runs on the 1-GPU box and
    covers everything but RCCL itself -- partition, windowed binning / render / backward, gather of the three channel segments
    into the root's tensor, all-reduce of the gradient records between the two halves of the backward, work-balanced
    re-partition, the refusal of render_depth;
  * two ranks on two GPUs, nccl (= RCCL) backend: the same worker, skipped unless the box has two devices.
Sharded frame vs the unsharded HIP run: image bit for bit, gradients <= 1e-5 of the largest entry."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import conftest

pytestmark = pytest.mark.gpu
WORKER = os.path.join(conftest.ROOT, "tests", "_shard_gpu_worker.py")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _spawn(backend, world=2):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, WORKER, backend], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert "SHARD_GPU_OK" in outs[0]


def test_two_ranks_on_one_gpu_host_staged_exchange():
    _spawn("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL between them)")
def test_two_ranks_two_gpus_rccl():
    _spawn("nccl")
