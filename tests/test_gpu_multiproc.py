"""One process per rank, as the driver launches bench.py (python -m torch.distributed.run --nproc-per-node N), with the real HIP
renderer in every rank: tiles dealt to 4 ranks, shards packed on the device, gathered, unpacked on rank 0 -- and the frame rank 0
then holds is the single-rank frame bit for bit, with and without the hit tiles.  The transport is gloo and every rank renders on
device 0 (a one-GPU box cannot give RCCL one device per rank); the RCCL call pattern itself runs at world 1 in test_gpu_parity."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,with_hits", [(4, False), (3, True)])
def test_ranks_as_processes_assemble_the_single_rank_frame(native_lib, world, with_hits):
    env = dict(os.environ, MOONRT_DIST_BACKEND="gloo", MOONRT_ONE_DEVICE="1", MP_WITH_HITS="1" if with_hits else "0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(HERE, "mp_gather_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert f"MP_GATHER OK world {world}" in p.stdout, tail


def test_ranks_on_their_own_gpus_over_rccl(native_lib):
    """The production transport: one device per rank, torch.distributed backend "nccl" (= RCCL), device-to-device gather.  Needs two
    GPUs in one box -- the one-GPU boxes this suite normally runs on skip it (counting devices does not initialise the GPU)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU(s) visible: RCCL needs one device per rank")
    world = min(n, 4)
    env = dict(os.environ, MP_WITH_HITS="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    env.pop("MOONRT_DIST_BACKEND", None); env.pop("MOONRT_ONE_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(HERE, "mp_gather_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert f"MP_GATHER OK world {world}" in p.stdout, tail
