"""Per-image data parallelism: one process per GPU, images are independent units,
weights replicated, no data-path collective.  The only communication is after the timed region:
max-over-ranks of the elapsed time and one gather of the per-image result rows (RCCL on GPUs - backend
"nccl" - or gloo on CPU)."""
from __future__ import annotations

import torch


def shard_image_ids(n_images: int, rank: int, world: int) -> list:
    """Static round-robin: image i -> rank i mod world (all images cost the same; scripts/tests.py:390)."""
    return list(range(rank, n_images, world))


def gather_results(elapsed_s: float, rows, device, failed_ids=None):
    """rows: list of (image_id, value...) floats of this rank.  Returns (max elapsed over ranks,
    float64 tensor of all ranks' rows, rank-major).  Works without an initialised process group.
    failed_ids (optional): ids of the images this rank could not process (a frame that raised);
    when given, a third value is returned: the sorted ids of all ranks' failures, so that rank 0 can report
    which images are missing from the table instead of silently averaging over fewer."""
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    r = (torch.tensor(rows, dtype=torch.float64, device=device).reshape(len(rows), -1) if len(rows)
         else torch.zeros(0, 2, dtype=torch.float64, device=device))
    if failed_ids is not None:
        t2, ftab = gather_results(elapsed_s, [(float(i), 0.0) for i in failed_ids], device)
        t3, table = gather_results(elapsed_s, rows, device)
        return max(t2, t3), table, sorted(int(v) for v in ftab[:, 0].tolist()) if ftab.numel() else []
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(t.item()), r.cpu()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, torch.tensor([r.shape[0]], dtype=torch.int64, device=device))
    width = r.shape[1] if r.numel() else 2
    nmax = int(max(int(c.item()) for c in counts))
    pad = torch.zeros(nmax, width, dtype=torch.float64, device=device)
    pad[:r.shape[0]] = r
    parts = [torch.empty_like(pad) for _ in counts]
    dist.all_gather(parts, pad)
    table = torch.cat([p[:int(c.item())] for p, c in zip(parts, counts)])
    return float(t.item()), table.cpu()
