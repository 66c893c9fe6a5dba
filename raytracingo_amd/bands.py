"""Row-band tiling of the framebuffer across ranks and the gather of the bands.

Rank g of G owns the window rows r with (r // band_h) % G == g, stored compactly (its k-th owned row is local row k).
Pixels are independent (seed = tea<16>(W*y + x, frame), kernel.cu:203-204), so there is no data-path exchange while
rendering; the only collective is one gather of the finished bands per presented frame.  With the "nccl" backend this
is RCCL over xGMI: a gather to one root lets the root ingest from its 7 links concurrently (no ring).
Plumbing only: torch.distributed moves bytes, nothing here computes pixels.
"""
import numpy as np


def band_rows(h, band_h, n_ranks, rank):
    """window rows owned by `rank`, in local (compact) order"""
    r = np.arange(h)
    return r[(r // band_h) % n_ranks == rank]


def max_local_rows(h, band_h, n_ranks):
    return max(len(band_rows(h, band_h, n_ranks, g)) for g in range(n_ranks))


def gather_bands(local, h, band_h, dist, dst=0, group=None, out=None, row_index=None, workspace=None):
    """Gather every rank's compact band tensor [rows_g, w, c] to `dst` and de-interleave into [h, w, c].

    `local` must be padded to max_local_rows (equal shapes on every rank).  Returns the full tensor on dst, None elsewhere.
    `row_index` (precomputed with full_row_index) and `workspace` (a [world * rows_pad, w, c] tensor on dst) let a per-frame
    caller avoid rebuilding the permutation and re-allocating the receive buffers."""
    import torch
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and out is None and workspace is None:
        return local[:h]
    if rank == dst:
        rows_pad = local.shape[0]
        stacked = workspace if workspace is not None else torch.empty((world * rows_pad,) + tuple(local.shape[1:]),
                                                                      dtype=local.dtype, device=local.device)
        parts = [stacked[g * rows_pad:(g + 1) * rows_pad] for g in range(world)]   # views: the receives land in place
        dist.gather(local, parts, dst=dst, group=group)
        if row_index is None:
            row_index = full_row_index(h, band_h, world, local.shape[0], local.device)
        if out is None:
            return stacked.index_select(0, row_index)
        torch.index_select(stacked, 0, row_index, out=out)
        return out
    dist.gather(local, None, dst=dst, group=group)
    return None


def full_row_index(h, band_h, n_ranks, padded_rows, device=None):
    """index into the concatenation of the padded per-rank buffers for each window row 0..h-1"""
    import torch
    idx = np.empty(h, dtype=np.int64)
    for g in range(n_ranks):
        rows = band_rows(h, band_h, n_ranks, g)
        idx[rows] = g * padded_rows + np.arange(len(rows))
    t = torch.from_numpy(idx)
    return t.to(device) if device is not None else t
