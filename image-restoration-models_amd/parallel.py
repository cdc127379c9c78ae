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
    which images are missing from the table instead of silently averaging over fewer.

    Collectives (identical on every rank whatever it passes, so ranks with and without rows or failures
    cannot desynchronise): ONE all_gather of (elapsed, rows, row width, failures) per rank, then ONE
    all_gather of the rows and failed ids padded to the largest count / width.  A rank without rows
    adopts the width of the others; ranks whose rows differ in width raise ValueError (on every rank)."""
    import torch.distributed as dist
    rows = [tuple(float(v) for v in r) for r in rows]
    width = len(rows[0]) if rows else 0
    if any(len(r) != width for r in rows):
        raise ValueError("gather_results: rows of one rank differ in length")
    fails = [float(i) for i in (failed_ids or [])]
    single = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1
    if single:
        table = torch.tensor(rows, dtype=torch.float64).reshape(len(rows), max(width, 2) if not rows else width)
        out = (float(elapsed_s), table)
        return out + (sorted(int(v) for v in fails),) if failed_ids is not None else out
    world = dist.get_world_size()
    meta = torch.tensor([elapsed_s, len(rows), width, len(fails)], dtype=torch.float64, device=device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = torch.stack(metas).cpu()
    counts = [int(v) for v in metas[:, 1].tolist()]
    widths = sorted({int(w) for w, n in zip(metas[:, 2].tolist(), counts) if n})
    nfail = [int(v) for v in metas[:, 3].tolist()]
    if len(widths) > 1:
        raise ValueError(f"gather_results: ranks disagree on the row width {widths}")
    wmax = widths[0] if widths else 2
    nmax, fmax = max(counts), max(nfail)
    payload = torch.zeros(nmax * wmax + fmax, dtype=torch.float64, device=device)
    if rows:
        payload[:len(rows) * wmax] = torch.tensor(rows, dtype=torch.float64, device=device).reshape(-1)
    if fails:
        payload[nmax * wmax:nmax * wmax + len(fails)] = torch.tensor(fails, dtype=torch.float64, device=device)
    parts = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(parts, payload)
    parts = [p.cpu() for p in parts]
    table = torch.cat([p[:n * wmax].reshape(n, wmax) for p, n in zip(parts, counts)])
    out = (float(metas[:, 0].max()), table)
    if failed_ids is None:
        return out
    allf = [int(v) for p, n in zip(parts, nfail) for v in p[nmax * wmax:nmax * wmax + n].tolist()]
    return out + (sorted(allf),)
