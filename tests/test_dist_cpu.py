"""World-size-2 gloo test (CPU) of the multi-process leg of bench.py: per-image sharding, max-over-ranks
timing and the single gather of (image id, PSNR) rows.  No GPU, no data-path collective."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, steps, q):
    sys.path.insert(0, ROOT)
    import irm_amd  # noqa: F401
    from irm_amd import parallel, synth
    from oracle import tiler_ref
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = parallel.shard_image_ids(world * steps, rank, world)
    rows = []
    for i in ids:                                   # the "model" is the identity: PSNR(input, target)
        inp, tgt = synth.synth_image_pair(i, 32, 48, 3, seed_base=1000, blur=5)
        rows.append((i, tiler_ref.psnr(tgt, inp)))
    elapsed = 0.01 * (rank + 1)
    tmax, table, failed = parallel.gather_results(elapsed, rows, torch.device("cpu"), failed_ids=[1000 + rank] if rank else [])
    if rank == 0:
        q.put((tmax, table.tolist(), failed))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, steps = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    tmax, table, failed = q.get()
    assert failed == [1001]                                         # rank 1's failed image id reaches rank 0
    assert abs(tmax - 0.02) < 1e-9                                  # MAX over ranks
    table = np.array(table)
    assert sorted(table[:, 0].astype(int).tolist()) == list(range(world * steps))   # every image exactly once
    sys.path.insert(0, ROOT)
    import irm_amd  # noqa: F401
    from irm_amd import synth
    from oracle import tiler_ref
    for i, p in table:
        inp, tgt = synth.synth_image_pair(int(i), 32, 48, 3, seed_base=1000, blur=5)
        assert abs(p - tiler_ref.psnr(tgt, inp)) < 1e-9


def test_shard_is_a_partition():
    sys.path.insert(0, ROOT)
    import irm_amd  # noqa: F401
    from irm_amd import parallel
    for n, w in [(64, 8), (7, 4), (3, 8), (16, 1)]:
        got = sorted(i for r in range(w) for i in parallel.shard_image_ids(n, r, w))
        assert got == list(range(n))
        assert all(i % w == r for r in range(w) for i in parallel.shard_image_ids(n, r, w))
