"""bench.py's multi-rank path on CPU: world_size 2, gloo.  The ranks run bench.BenchLoop itself (step / drain / reduce /
per_rank and the seed partition) and the real ``Ensemble.gather_trajectories``; only the engine underneath — which needs a
GPU — is a stand-in that returns a deterministic trajectory block per (rank, step).  Also checks how ``bench.py --gpus N``
treats its launcher environment."""
import os
import socket
import subprocess
import sys
import textwrap

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
    import bench
    from vgsim_amd.ensemble import Ensemble, EnsembleResult
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R, T, P, N = 3, 5, 2, 40

    def block(seeds):            # what this rank's engine "simulated": a function of its seeds only
        return seeds[:, None, None, None].astype(np.float64) + np.arange(T * P * 2, dtype=np.float64).reshape(1, T, P, 2) * 100000.0

    class StubEnsemble(Ensemble):     # engine stand-in: everything above the C ABI is the product's code
        def __init__(self):
            self.R, self.traj_shape, self.calls = R, (R, T, P, 2), []
            self.model = type("Model", (), {"sizes": np.array([10 ** 7, 10 ** 7])})   # the 32-bit wire format asks for the sizes
        def simulate(self, iterations, sample_size=None, record_events=False, traj_points=0, traj_window=(0.0, 1.0), seeds=None, **kw):
            assert iterations == N and traj_points == T and len(seeds) == R
            self.calls.append(np.asarray(seeds).copy())
            self._block = block(np.asarray(seeds))
            res = EnsembleResult(R)
            res.events[:] = N + rank          # ranks differ so the SUM is checkable
            res.kernel_ms = 1.0
            return res
        def trajectories(self, out=None):
            return self._block.copy()

    ens = StubEnsemble()
    loop = bench.BenchLoop(ens, R, N, T, world=world, rank=rank, device="cpu", pops=P)
    events = 0
    for i in range(3):
        events += loop.step(i).total_events     # posts step i's gather; step i+1 waits for it before posting its own
    loop.drain()
    if rank == 0:
        want = np.stack([block(2020 + (2 * world + k) * R + np.arange(R)) for k in range(world)])
        assert loop.gather_out.shape == (world, R, T, P, 2)
        assert loop.gather_out.dtype == torch.int32     # whole numbers below the population size: 32 bits on the wire
        assert np.array_equal(loop.gather_out.numpy(), want), "gathered trajectories of the last step"
    else:
        assert loop.gather_out is None
    # seeds: disjoint across (step, rank), independent of how many ranks there are
    mine = torch.from_numpy(np.concatenate(ens.calls))
    allseeds = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allseeds, mine)
    flat = torch.cat(allseeds).numpy()
    assert len(set(flat.tolist())) == 3 * world * R and flat.min() == 2020 and flat.max() == 2020 + 3 * world * R - 1
    elapsed, total = loop.reduce(1.0 + rank, events)
    assert elapsed == float(world) and total == sum(3 * R * (N + k) for k in range(world)), (elapsed, total)
    assert loop.per_rank(events) == [float(3 * R * (N + k)) for k in range(world)]
    if rank == 0:
        print("BENCHLOOP_OK")
    dist.destroy_process_group()
""")


def test_bench_loop_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "BENCHLOOP_OK" in outs[0]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """Under a launcher the rank count must equal --gpus: no silent one-rank run reported as N GPUs."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and '"metric"' not in r.stdout


def test_bench_gpus_n_starts_n_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (a child torch.distributed.run); on a box
    without two GPUs the ranks fail and the parent exits non-zero without printing a result line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "2-rank launch failed" in r.stderr


WORKER8 = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    import bench
    from vgsim_amd.ensemble import Ensemble, EnsembleResult
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, P, N = 1001, bench.POPS, 100000

    class StubEnsemble(Ensemble):     # engine stand-in: partition, error vote, gather and reductions are bench.py's / the product's own
        def __init__(self, R):
            self.R, self.traj_shape, self.seeds_seen = R, (R, T, P, 2), None
            self.model = type("Model", (), {"sizes": np.full(P, 10 ** 7)})
        def simulate(self, iterations, sample_size=None, record_events=False, traj_points=0, traj_window=(0.0, 1.0), seeds=None, **kw):
            assert iterations == N and traj_points == T and len(seeds) == self.R
            self.seeds_seen = np.asarray(seeds).copy()
            res = EnsembleResult(self.R)
            res.events[:] = N
            res.kernel_ms = 1.0
            return res
        def trajectories(self, out=None):      # replicate r's block carries its seed: the gathered result says who sent what
            return np.broadcast_to(self.seeds_seen[:, None, None, None].astype(np.float64), self.traj_shape).copy()
        def close(self):
            pass

    R, seeds = bench.config5_partition(world, rank)
    assert R == 32 and seeds[0] == 2020 + 32 * rank and seeds[-1] == 2020 + 32 * rank + 31
    ens = StubEnsemble(R)
    o = bench.config5_leg("cpu", world=world, rank=rank, events=N, traj_points=T, ens=ens)
    assert o["replicates_per_gpu"] == 32 and o["gather_bytes_per_gpu"] == 32 * 1001 * 64 * 2 * 8
    assert abs(o["value"] * (o["simulate_s"] + 1e-3 * o["gather_ms"]) - 256 * N) / (256 * N) < 0.5     # all ranks' events over the slowest rank's time
    if rank == 0:
        assert o["gathered_shape_on_rank0"] == [8, 32, 1001, 64, 2], o["gathered_shape_on_rank0"]
        out = ens.gather_trajectories(dst=0)
        got = out[:, :, 0, 0, 0].numpy().reshape(-1)
        assert np.array_equal(got, 2020.0 + np.arange(256)), "rank k's block sits in row k, seeds 2020..2275 in order"
        # the headline leg's memory plan of rank 0 at eight ranks on a 288 GB device: must fit (and say so) before anything is allocated
        bench.memory_plan(16384, 100000, 1001, 8, 0, int(280e9), int(288e9))
        print("CONFIG5_OK")
    else:
        assert o["gathered_shape_on_rank0"] is None
        ens.gather_trajectories(dst=0)
    dist.destroy_process_group()
""")


def test_config5_partition_and_gather_world_size_8(tmp_path):
    """BASELINE config 5 as the driver will run it on a whole node — eight ranks, 32 replicates each, one gather of
    [8, 32, 1001, 64, 2] to rank 0 — on eight CPU ranks (gloo) with an engine stand-in: bench.config5_leg's own partition,
    error vote, gather and reductions, and rank 0's memory plan at 288 GB."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % ROOT)
    port = free_port()
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "CONFIG5_OK" in outs[0]
