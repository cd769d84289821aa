"""Multi-process path on CPU: world_size 2, gloo.  The engine itself needs a GPU, so the per-rank trajectory
block is a stand-in array here; what is exercised is the one collective of the ensemble layer (rank order,
shapes, dst-only result) and the seed partition that makes results independent of the GPU count."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from vgsim_amd.ensemble import Ensemble
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R, T, P = 3, 5, 2
    ens = Ensemble.__new__(Ensemble)          # no GPU here: bypass engine creation
    ens.R, ens.traj_shape = R, (R, T, P, 2)
    block = (np.arange(R * T * P * 2, dtype=np.float64).reshape(R, T, P, 2) + 1000.0 * rank)
    ens.trajectories = lambda out=None: block
    got = ens.gather_trajectories(dst=0)
    if rank == 0:
        assert got.shape == (world, R, T, P, 2)
        for k in range(world):
            assert np.array_equal(got[k].numpy(), np.arange(R * T * P * 2, dtype=np.float64).reshape(R, T, P, 2) + 1000.0 * k)
        print("GATHER_OK")
    else:
        assert got is None
    # asynchronous form used by bench.py: the transfer of step i overlaps step i+1; a preallocated result is reused
    out = torch.empty((world, R, T, P, 2), dtype=torch.float64) if rank == 0 else None
    pend = None
    for it in range(3):
        block = (np.arange(R * T * P * 2, dtype=np.float64).reshape(R, T, P, 2) + 1000.0 * rank + 7.0 * it)
        ens.trajectories = lambda out=None, b=block: b.copy()
        if pend is not None:
            res = pend.wait()
            if rank == 0:
                assert res is out and np.array_equal(res[1].numpy(), np.arange(R * T * P * 2, dtype=np.float64).reshape(R, T, P, 2) + 1000.0 + 7.0 * (it - 1))
        pend = ens.gather_trajectories(dst=0, out=out, async_op=True)
    res = pend.wait()
    if rank == 0:
        assert np.array_equal(res[0].numpy(), np.arange(R * T * P * 2, dtype=np.float64).reshape(R, T, P, 2) + 14.0)
        print("ASYNC_OK")
    # 32-bit wire format: same numbers, int32 result; refused when a population could overflow it
    ens.model = type("Model", (), {"sizes": np.array([10 ** 7, 10 ** 7])})
    got32 = ens.gather_trajectories(dst=0, wire_dtype=torch.int32)
    if rank == 0:
        assert got32.dtype == torch.int32 and np.array_equal(got32.numpy(), res.numpy())
        print("WIRE32_OK")
    ens.model = type("Model", (), {"sizes": np.array([10 ** 7, 2 ** 31])})
    try:
        ens.gather_trajectories(dst=0, wire_dtype=torch.int32)
        raise SystemExit("int32 wire format accepted a population of 2^31 hosts")
    except ValueError:
        pass
    # seed partition used by bench.py: disjoint and independent of the world size
    step, seeds = 0, 2020 + (0 * world + rank) * R + np.arange(R)
    allseeds = [torch.zeros(R, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allseeds, torch.from_numpy(seeds))
    flat = torch.cat(allseeds).numpy()
    assert len(set(flat.tolist())) == world * R and flat.min() == 2020 and flat.max() == 2020 + world * R - 1
    dist.destroy_process_group()
""")


def test_gather_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0] and "ASYNC_OK" in outs[0] and "WIRE32_OK" in outs[0]
