"""Frame-parallel sharding across the GPUs of a node: frame f -> rank f mod G, no exchange.

torch.distributed is used for the control plane only (barrier, max over ranks of a timing); the data
path has no collective.  Pure host logic -- exercised by world_size-2 gloo tests on CPU.

This module imports nothing that touches a GPU: `spawn_ranks` must be callable from a parent process
that has not initialised HIP (a process that has may neither fork safely nor exec).
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    """A TCP port nobody listens on right now, on the loopback interface the rendezvous uses."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    """Environment of rank `rank` of a one-node, one-process-per-GPU job (what torch.distributed.run sets)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(world, argv, timeout_s=None, poll_s=0.05):
    """Start `world` fresh child processes of `argv` (rank r gets rank_env(r)), wait for all of them and return
    the list of exit codes.  Never an exec of the caller: children are ordinary subprocesses, stdout / stderr
    inherited (rank 0 prints the job's one JSON line).  As soon as one child fails -- or the timeout expires --
    the remaining children (exactly the PIDs started here) are terminated, so that a rank waiting in a barrier
    for a dead peer cannot hang the job; their exit code is then reported as non-zero too."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = [subprocess.Popen(list(argv), env=rank_env(r, world, port)) for r in range(world)]
    t0 = time.monotonic()
    failed = False
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes) or (timeout_s is not None and time.monotonic() - t0 > timeout_s):
            failed = True
            break
        time.sleep(poll_s)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    codes = [p.returncode for p in procs]
    if failed and all(c == 0 for c in codes):  # timeout with every child exiting 0 at the last moment
        codes[0] = 124
    return codes


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


def gather_objects(dist, obj):
    """Every rank contributes one picklable object; returns the list over ranks, in rank order, on every rank
    (bench.py: the per-rank line -- device, PCI id, rate, kernel time, engine clock -- so that a straggling GPU of an
    N-GPU run is visible in rank 0's one JSON line, which otherwise carries only the max elapsed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, obj)
    return parts


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


# ---- host topology: which CPUs sit next to which GPU (no HIP call: usable before the runtime is initialised) ----------

def parse_cpulist(text):
    """'0-63,128-191' -> {0..63, 128..191} (the format of sysfs cpulist files)."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def gpu_pci_ids(sysfs_root="/sys"):
    """PCI ids ('0000:c1:00.0') of the node's GPUs in the order the HIP runtime numbers them when no *_VISIBLE_DEVICES
    variable filters them: the KFD topology nodes with SIMDs, in node order (location_id = bus << 8 | device << 3 | function)."""
    base = os.path.join(sysfs_root, "class", "kfd", "kfd", "topology", "nodes")
    ids = []
    try:
        nodes = sorted((n for n in os.listdir(base) if n.isdigit()), key=int)
    except OSError:
        return ids
    for n in nodes:
        props = {}
        try:
            with open(os.path.join(base, n, "properties")) as fh:
                for line in fh:
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
        except OSError:
            continue
        if int(props.get("simd_count", "0")) == 0:
            continue  # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        ids.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return ids


def visible_device_index(local_index, env=None):
    """The physical GPU behind HIP device `local_index` when HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES hold plain
    index lists (anything else -- UUIDs -- is left alone: None)."""
    env = os.environ if env is None else env
    idx = local_index
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):  # HIP's list indexes into ROCR's
        v = env.get(var)
        if v is None or v == "":
            continue
        try:
            lst = [int(x) for x in v.split(",")]
        except ValueError:
            return None
        if not 0 <= idx < len(lst):
            return None
        idx = lst[idx]
    return idx


def cpus_near_gpu(local_index, sysfs_root="/sys", env=None):
    """CPUs on the socket / NUMA node the GPU's host bridge hangs off (sysfs local_cpulist of its PCI function), or an
    empty set when the topology cannot be read.  The page-locked staging memory of a rank lives in that node's DRAM, so a
    rank that runs there copies locally and its DMA does not cross the socket link (host DRAM is what the 8
    ranks of a node share)."""
    phys = visible_device_index(local_index, env)
    ids = gpu_pci_ids(sysfs_root)
    if phys is None or not 0 <= phys < len(ids):
        return set()
    try:
        with open(os.path.join(sysfs_root, "bus", "pci", "devices", ids[phys], "local_cpulist")) as fh:
            return parse_cpulist(fh.read())
    except (OSError, ValueError):
        return set()


def pin_to_gpu_cpus(local_index, sysfs_root="/sys"):
    """Restrict this process to the CPUs near its GPU (intersected with what it may already use).  Call before the first HIP
    call so that the runtime's helper threads inherit the mask.  Returns the sorted CPU list applied, or [] when nothing
    was changed."""
    near = cpus_near_gpu(local_index, sysfs_root)
    if not near or not hasattr(os, "sched_setaffinity"):
        return []
    pick = os.sched_getaffinity(0) & near
    if not pick:
        return []
    os.sched_setaffinity(0, pick)
    return sorted(pick)


# ---- one seeded frame set over N ranks, and "N-GPU output == 1-GPU output"  ---------------------------

def global_frames_of_rank(frames_per_rank, rank, world):
    """Weak scaling on ONE frame set: the job's frames 0 .. frames_per_rank * world - 1 are dealt f mod world, so every rank
    holds frames_per_rank of them and a 1-rank job holds exactly the frames rank 0 .. world-1 would share."""
    return frames_of_rank(frames_per_rank * world, rank, world)


def merge_frame_hashes(dist, local):
    """{global frame index: sha256 hex} of the frames each rank hashed -> the union on every rank.  A frame reported by two
    ranks is an error (the deal is a partition); frames nobody hashed are simply absent (ranks hash a sample)."""
    merged = {}
    for part in gather_objects(dist, dict(local)):
        overlap = set(merged) & set(part)
        if overlap:
            raise RuntimeError("frames filtered by two ranks: %s" % sorted(overlap))
        merged.update(part)
    return merged


def frames_equal_across_ranks(dist, rank, world, local_hashes, refilter, per_rank=2):
    """Rank 0 filters `per_rank` of the frames every OTHER rank hashed on its own GPU again (refilter(global index) -> sha256
    hex of the filtered frame) and compares: the N-GPU job's output equals what one GPU produces, frame by frame, on the
    sample.  Returns the same dict on every rank: frames_hashed, frames_rechecked, frames_equal_1gpu, mismatches."""
    merged = merge_frame_hashes(dist, local_hashes)
    res = None
    if rank == 0:
        mism, n = [], 0
        for r in range(1, world):
            theirs = sorted(g for g in merged if owner_of_frame(g, world) == r)[:per_rank]
            for g in theirs:
                n += 1
                if refilter(g) != merged[g]:
                    mism.append(g)
        res = {"frames_hashed": len(merged), "frames_rechecked_on_rank0": n, "frames_equal_1gpu": not mism, "mismatches": mism}
    return gather_objects(dist, res)[0]


def aggregate_rate(per_rank, units_key, seconds_key):
    """Whole-job rate of a leg every rank ran between the same two barriers: the units all ranks processed / the slowest
    rank's time (the contract's max-over-ranks)."""
    secs = [r[seconds_key] for r in per_rank if r.get(seconds_key)]
    if not secs:
        return None
    return sum(r[units_key] for r in per_rank if r.get(seconds_key)) / max(secs)
