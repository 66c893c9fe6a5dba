"""The N>1 path on CPU: world_size-2 (and 3) gloo groups run the SAME band partition + gather code bench.py uses
(raytracingo_amd/bands.py); each rank's band is rendered by the oracle standing in for the GPU, and rank 0 must end up
with the full oracle image bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, H, N, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from raytracingo_amd import bands
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = O.scene("cornell", W, H)
        band_h = 4
        acc, img, c = O.render(sc, O.frame(W, H, N, 0, path=True, bands=(band_h, world, rank), mode=1, threads=1))
        rows_pad = bands.max_local_rows(H, band_h, world)
        assert acc.shape[0] == len(bands.band_rows(H, band_h, world, rank))
        for arr, name in ((acc, "accum"), (img, "image")):
            local = torch.zeros((rows_pad,) + arr.shape[1:], dtype=torch.from_numpy(arr).dtype)
            local[:arr.shape[0]] = torch.from_numpy(arr)
            full = bands.gather_bands(local, H, band_h, dist, dst=0)
            if rank == 0:
                np.save(os.path.join(tmp, name + ".npy"), full.numpy())
            else:
                assert full is None
        rays = torch.tensor([c["rays_total"]], dtype=torch.int64)
        dist.all_reduce(rays)
        if rank == 0:
            np.save(os.path.join(tmp, "rays.npy"), rays.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_gather_reassembles_the_frame(tmp_path, world, oracle):
    W, H, N = 64, 38, 1   # 38 rows: a ragged last band
    port = 29500 + world + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, W, H, N, str(tmp_path)), nprocs=world, join=True)
    sc = oracle.scene("cornell", W, H)
    acc, img, c = oracle.render(sc, oracle.frame(W, H, N, 0, path=True, mode=1))
    got = np.load(os.path.join(str(tmp_path), "accum.npy"))
    assert np.array_equal(got.view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "image.npy")), img)
    assert int(np.load(os.path.join(str(tmp_path), "rays.npy"))[0]) == c["rays_total"]


def test_band_bookkeeping():
    from raytracingo_amd import bands
    for h, b, g in [(1080, 4, 8), (2160, 4, 8), (38, 4, 3), (7, 4, 2), (3, 4, 8)]:
        rows = np.concatenate([bands.band_rows(h, b, g, r) for r in range(g)])
        assert sorted(rows.tolist()) == list(range(h))
        pad = bands.max_local_rows(h, b, g)
        idx = bands.full_row_index(h, b, g, pad).numpy()
        assert len(set(idx.tolist())) == h and idx.max() < g * pad
    # 1080 rows over 8 GPUs in 4-row bands: 270 bands -> 34 or 33 bands per GPU (load balance within 3 %)
    counts = [len(bands.band_rows(1080, 4, 8, r)) for r in range(8)]
    assert max(counts) - min(counts) <= 4
