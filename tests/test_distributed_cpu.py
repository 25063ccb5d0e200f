"""The N>1 path on CPU: world_size-2 gloo.  Each rank produces the per-pixel RGB *sum* of
its sample shard (here with the oracle standing in for the GPU, same C-ABI parameters:
sample_begin / sample_count / output_sum), the sums are reduced to rank 0 and divided by
spp exactly as bench.py does on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from crucible_amd.distributed import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,spp", [(1, 512), (2, 512), (4, 512), (8, 512), (3, 10), (8, 5), (2, 1)])
def test_shards_partition_the_samples(world, spp):
    covered = []
    for r in range(world):
        b, n = shard_range(r, world, spp)
        covered.extend(range(b, b + n))
    assert covered == list(range(spp))
    sizes = [shard_range(r, world, spp)[1] for r in range(world)]
    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(world, world, spp)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from crucible_amd import _abi as A
    from crucible_amd.demo_builder import book1_end_scene
    from crucible_amd.distributed import reduce_to_mean, shard_range
    from oracle.oracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = book1_end_scene(1, scene_seed=1, image_width=48, samples=6)
        cam = sc.scene_cam
        o = Oracle(A.CR_REAL_F32)
        b, n = shard_range(rank, world, cam.samples)
        sums, _ = o.render_image(sc, seed=9, sample_begin=b, sample_count=n, output_sum=True, n_threads=2)
        t = torch.from_numpy(sums.copy())
        reduce_to_mean(t, cam.samples, dst=0)
        if rank == 0:
            full, _ = o.render_image(sc, seed=9, n_threads=2)
            np.savez(out_path, reduced=t.numpy(), full=full)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_world2_reduce_matches_single_rank(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    z = np.load(out)
    # identical sample set; only the order of the f32 adds differs
    assert np.abs(z["reduced"] - z["full"]).max() <= 1e-6
    assert z["reduced"].shape == (27, 48, 3)


def test_reduce_to_mean_without_process_group():
    import torch
    from crucible_amd.distributed import reduce_to_mean
    t = torch.tensor([2.0, 4.0, 8.0])
    assert reduce_to_mean(t, 4).tolist() == [0.5, 1.0, 2.0]
