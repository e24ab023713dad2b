"""The N > 1 path on CPU: world_size-2 `gloo` run of the same sharding + reduce code bench.py
uses on RCCL (path_tracer_amd.distributed).  Each rank's shard is rendered by the oracle
(no GPU here); the reduced frame on rank 0 must agree with the unsharded frame to fp32
summation-order rounding, and each shard must be what the single-process run computes."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch
    import oracle_lib
    from conftest import RENDER_SEED, make_raytracer
    from path_tracer_amd import distributed as ptdist
    assert ptdist.env_world() == (rank, rank, world)
    dist = ptdist.init_process_group("gloo")
    samples = 3
    rt = make_raytracer("rtow", 40, 24, samples)
    osc = oracle_lib.OracleScene(rt.packed)
    s0, s1 = ptdist.shard_range(samples, rank, world)
    part = osc.render(rt.camera.pod, samples, seed=RENDER_SEED, sample_range=(s0, s1), threads=2)["radiance"]
    np.save(os.path.join(out_dir, "part%d.npy" % rank), part)
    accum = torch.from_numpy(part.copy()).reshape(-1)
    ptdist.reduce_radiance(accum, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), accum.numpy().reshape(part.shape))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_reduce(tmp_path, oracle):
    import torch.multiprocessing as mp
    from conftest import RENDER_SEED, make_raytracer
    from path_tracer_amd import distributed as ptdist
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    samples = 3
    rt = make_raytracer("rtow", 40, 24, samples)
    osc = oracle.OracleScene(rt.packed)
    full = osc.render(rt.camera.pod, samples, seed=RENDER_SEED)["radiance"]
    parts = [np.load(tmp_path / ("part%d.npy" % r)) for r in range(world)]
    reduced = np.load(tmp_path / "reduced.npy")
    for r in range(world):
        want = osc.render(rt.camera.pod, samples, seed=RENDER_SEED, sample_range=ptdist.shard_range(samples, r, world))
        assert np.array_equal(parts[r], want["radiance"])
    assert np.array_equal(reduced, parts[0] + parts[1])          # the reduce is a plain fp32 sum
    assert np.allclose(reduced, full, rtol=2e-6, atol=1e-6)      # vs the sequential order: rounding only
    a, b = osc.resolve(reduced, samples).astype(int), osc.resolve(full, samples).astype(int)
    assert np.mean(np.all(np.abs(a - b) <= 1, axis=-1)) >= 0.999  # north_star tolerance
