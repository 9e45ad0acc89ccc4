"""Frame-parallel sharding across the GPUs of a node: frame f -> rank f mod G, no exchange.

torch.distributed is used for the control plane only (barrier, max over ranks of a timing); the data
path has no collective.  Pure host logic -- exercised by world_size-2 gloo tests on CPU.
"""


def frames_of_rank(n_frames, rank, world):
    """Indices of the frames rank `rank` filters when n_frames are dealt round-robin over `world` GPUs."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return list(range(rank, n_frames, world))


def owner_of_frame(frame, world):
    return frame % world


def max_over_ranks(dist, value):
    """Max of a python float over all ranks (dist = torch.distributed module or None for one process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_frame_results(dist, local, n_frames):
    """Test/verification helper: every rank contributes {frame_index: sha256 hex}; returns the merged
    dict on every rank.  Used to check that an N-GPU run equals the 1-GPU run frame by frame."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        merged = dict(local)
    else:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, dict(local))
        merged = {}
        for p in parts:
            overlap = set(merged) & set(p)
            if overlap:
                raise RuntimeError("frames filtered by two ranks: %s" % sorted(overlap))
            merged.update(p)
    missing = set(range(n_frames)) - set(merged)
    if missing:
        raise RuntimeError("frames filtered by no rank: %s" % sorted(missing))
    return merged
