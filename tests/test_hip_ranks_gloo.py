"""Two ranks with REAL engines (both on the one GPU of the test box, gloo for the collective): bench.BenchLoop's step / drain /
reduce over ``Ensemble.simulate`` and ``Ensemble.gather_trajectories`` — the path the 8-GPU scaling run takes with RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    import bench
    from vgsim_amd.ensemble import Ensemble
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R, N, T = 2048, 300, 9
    ens = Ensemble(bench.make_simulator(2020), R, device=0)
    loop = bench.BenchLoop(ens, R, N, T, world=world, rank=rank, device="cpu")
    events = 0
    for i in range(2):
        events += loop.step(i).total_events
    loop.drain()
    elapsed, total = loop.reduce(1.0, events)
    assert total == 2 * world * R * N, total
    if rank == 0:
        got = loop.gather_out.numpy()
        assert got.shape == (world, R, T, bench.POPS, 2)
        # rank 0's own block equals what its engine holds; rank 1's block differs (other seeds) and is a valid trajectory
        assert np.array_equal(got[0], ens.trajectories())
        assert not np.array_equal(got[0], got[1])
        hosts = got[1].sum(axis=(2, 3))                     # infectious + susceptible over all populations: conserved
        assert (hosts == hosts[0, 0]).all()
        print("RANKS_OK")
    ens.close()
    dist.destroy_process_group()
""")


def test_two_ranks_real_engines_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RANKS_OK" in outs[0]
