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


def _worker_ragged(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import irm_amd  # noqa: F401
    from irm_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 0: three rows of width 4 (id, psnr, ssim, ms), no failed_ids argument at all; rank 1: every frame failed
    if rank == 0:
        res = parallel.gather_results(0.5, [(i, 30.0 + i, 0.9, 12.5) for i in range(3)], torch.device("cpu"))
    else:
        res = parallel.gather_results(0.25, [], torch.device("cpu"), failed_ids=[7, 5])
    q.put((rank, res[0], res[1].tolist(), list(res[2]) if len(res) > 2 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_with_an_empty_rank_wider_rows_and_mixed_failed_ids():
    """a rank without rows must not fix the padded width at 2, and ranks that pass / do not pass
    failed_ids must still run the same collectives."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict((r[0], r[1:]) for r in (q.get(), q.get()))
    for rank in (0, 1):
        tmax, table, failed = got[rank]
        assert tmax == 0.5 and np.array(table).shape == (3, 4) and table[2] == [2.0, 32.0, 0.9, 12.5]
        assert failed == (None if rank == 0 else [5, 7])


def test_gather_single_process():
    sys.path.insert(0, ROOT)
    import irm_amd  # noqa: F401
    from irm_amd import parallel
    t, tab, failed = parallel.gather_results(1.5, [(0, 31.0), (1, 29.0)], torch.device("cpu"), failed_ids=[3])
    assert t == 1.5 and tab.tolist() == [[0.0, 31.0], [1.0, 29.0]] and failed == [3]
    t, tab = parallel.gather_results(1.5, [], torch.device("cpu"))
    assert tab.shape[0] == 0


def test_shard_is_a_partition():
    sys.path.insert(0, ROOT)
    import irm_amd  # noqa: F401
    from irm_amd import parallel
    for n, w in [(64, 8), (7, 4), (3, 8), (16, 1)]:
        got = sorted(i for r in range(w) for i in parallel.shard_image_ids(n, r, w))
        assert got == list(range(n))
        assert all(i % w == r for r in range(w) for i in parallel.shard_image_ids(n, r, w))
