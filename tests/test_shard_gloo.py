"""Multi-process (world_size 2, gloo, CPU) test of the frame-parallel sharding used for N GPUs:
every frame is owned by exactly one rank, the per-frame results merged over ranks equal the
single-process results, and the timing reduction is a max.  The per-frame 'filter' here is the
oracle (test infrastructure) standing in for a GPU: this test covers the distributed plumbing, the
GPU tests cover the kernel."""
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, n_frames, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gpu_video_codec_amd import shard, synth
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = {}
    for f in shard.frames_of_rank(n_frames, rank, world):
        y = synth.blocky_plane(64, 48, seed=3, frame=f)
        local[f] = hashlib.sha256(oracle.filter_plane(y, 32).tobytes()).hexdigest()
    merged = shard.gather_frame_results(dist, local, n_frames)
    tmax = shard.max_over_ranks(dist, 1.0 + rank)
    dist.barrier()
    dist.destroy_process_group()
    out_q.put((rank, merged, tmax))


def test_two_rank_frame_parallel_equals_single_process():
    import torch.multiprocessing as mp
    from gpu_video_codec_amd import shard, synth
    from oracle import oracle
    n_frames, world = 7, 2
    want = {f: hashlib.sha256(oracle.filter_plane(synth.blocky_plane(64, 48, seed=3, frame=f), 32).tobytes()).hexdigest()
            for f in range(n_frames)}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, merged, tmax in res:
        assert merged == want
        assert tmax == 2.0  # max over ranks of (1 + rank)


def test_shard_map_properties():
    from gpu_video_codec_amd import shard
    for n in (0, 1, 5, 64, 129):
        for g in (1, 2, 4, 8):
            parts = [shard.frames_of_rank(n, r, g) for r in range(g)]
            flat = sorted(x for p in parts for x in p)
            assert flat == list(range(n))
            assert all(shard.owner_of_frame(f, g) == r for r, p in enumerate(parts) for f in p)
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard.frames_of_rank(4, 2, 2)
    with pytest.raises(RuntimeError):
        shard.gather_frame_results(None, {0: "a"}, 2)
