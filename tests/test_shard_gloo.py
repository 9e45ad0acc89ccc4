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


# ---- the launcher of `python bench.py --gpus N` (shard.spawn_ranks) ------------------------------------

_CHILD_OK = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from gpu_video_codec_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
dist.barrier()
t = shard.max_over_ranks(dist, 10.0 + rank)
per_rank = shard.gather_objects(dist, {"rank": rank, "device": rank, "frames_per_s": 100.0 * (rank + 1)})
dist.barrier()
dist.destroy_process_group()
with open(os.path.join(%(out)r, "rank%%d.json" %% rank), "w") as fh:
    json.dump({"rank": rank, "world": world, "tmax": t, "pid": os.getpid(), "ppid": os.getppid(), "per_rank": per_rank}, fh)
"""

_CHILD_FAIL = r"""
import os, sys, time
if os.environ["RANK"] == "1":
    sys.exit(7)          # one rank dies before the rendezvous ...
time.sleep(600)          # ... the others would wait for it forever
"""


def test_spawn_ranks_starts_n_fresh_processes(tmp_path):
    import json
    from gpu_video_codec_amd import shard
    script = tmp_path / "child.py"
    script.write_text(_CHILD_OK % {"root": ROOT, "out": str(tmp_path)})
    codes = shard.spawn_ranks(3, [sys.executable, str(script)], timeout_s=240)
    assert codes == [0, 0, 0]
    got = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    assert [g["rank"] for g in got] == [0, 1, 2] and all(g["world"] == 3 for g in got)
    assert all(g["tmax"] == 12.0 for g in got)                     # max over ranks of 10 + rank
    # every rank holds every rank's line, in rank order (bench.py's `per_rank`: a straggling GPU is visible on rank 0)
    assert all(g["per_rank"] == [{"rank": r, "device": r, "frames_per_s": 100.0 * (r + 1)} for r in range(3)] for g in got)
    assert shard.gather_objects(None, {"rank": 0}) == [{"rank": 0}]
    assert len({g["pid"] for g in got}) == 3                       # three distinct processes ...
    assert all(g["ppid"] == os.getpid() and g["pid"] != os.getpid() for g in got)  # ... children, not an exec of the caller


def test_spawn_ranks_fails_when_a_rank_fails(tmp_path):
    import time
    from gpu_video_codec_amd import shard
    script = tmp_path / "child.py"
    script.write_text(_CHILD_FAIL)
    t0 = time.monotonic()
    codes = shard.spawn_ranks(2, [sys.executable, str(script)], timeout_s=120)
    assert time.monotonic() - t0 < 60        # the surviving rank was terminated, not waited for
    assert codes[1] == 7 and codes[0] != 0


def test_bench_gpus_flag_spawns_before_any_gpu_call(tmp_path, monkeypatch):
    """`python bench.py --gpus 2` with WORLD_SIZE unset must hand over to shard.spawn_ranks with the same
    arguments BEFORE the product library is loaded (no HIP call in the launcher); under torch.distributed.run
    (WORLD_SIZE set) it must not spawn again."""
    import importlib.util
    from gpu_video_codec_amd import _lib, shard
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    def fake_spawn(world, argv, **kw):
        calls.append((world, list(argv), kw))
        return [0] * world

    monkeypatch.setattr(shard, "spawn_ranks", fake_spawn)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    bench.main()
    assert len(calls) == 1
    world, argv, kw = calls[0]
    assert kw.get("timeout_s") == 540.0      # a rank stuck in a barrier is terminated, not waited for until the driver kills the job
    assert world == 2 and argv[0] == sys.executable and argv[1].endswith("bench.py")
    assert argv[2:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]

    def failing_spawn(world, argv, **kw):
        return [0, 3]

    monkeypatch.setattr(shard, "spawn_ranks", failing_spawn)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3

    # in a fresh interpreter: at the moment of the hand-over the product library (and with it the HIP runtime)
    # has not been loaded
    import subprocess
    probe = (
        "import sys, importlib.util; sys.argv = ['bench.py', '--gpus', '2']\n"
        "spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec)\n"
        "spec.loader.exec_module(b)\n"
        "from gpu_video_codec_amd import shard, _lib\n"
        "def fake(world, argv, **kw):\n"
        "    print('LOADED' if _lib._lib is not None else 'NOTLOADED'); return [0] * world\n"
        "shard.spawn_ranks = fake\n"
        "b.main()\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "NOTLOADED"


def test_clock_power_sampler_without_a_card():
    """bench.py wraps its timed region in tools/clock_power_trace.Sampler; with no readable amdgpu sysfs files (this
    container, or a box that hides them) it must cost nothing and report nothing."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from clock_power_trace import Sampler, _current_mhz
    s = Sampler(0, pci="ffff:ff:1f.7").start()   # a PCI address no card has: no HIP call is made
    r = s.stop()
    assert r["samples"] == 0 and r["engine_clock_MHz"] is None and r["socket_power_W"] is None
    assert _current_mhz("0: 132Mhz\n1: 2400Mhz *\n") == 2400.0 and _current_mhz("") is None


def test_bench_child_runner_kills_the_whole_process_group(tmp_path):
    """bench.py starts its helper measurements (copy floor, live PMC passes) as children in their OWN process group and
    kills the group on timeout: a grandchild (rocprofv3 -> python) must not survive holding the GPU."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pidfile = tmp_path / "grandchild.pid"
    child = ("import subprocess, sys, time\n"
             "g = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"
             "open(%r, 'w').write(str(g.pid))\n"
             "time.sleep(600)\n" % str(pidfile))
    t0 = time.monotonic()
    out, err, rc = bench.run_group([sys.executable, "-c", child], 3)
    assert rc is None and time.monotonic() - t0 < 30
    gpid = int(pidfile.read_text())
    for _ in range(50):
        try:
            os.kill(gpid, 0)
        except ProcessLookupError:
            break
        # a killed but not yet reaped grandchild shows as a zombie owned by init: state Z counts as gone
        try:
            if open("/proc/%d/stat" % gpid).read().split(") ")[1][0] == "Z":
                break
        except OSError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("grandchild %d survived the timeout" % gpid)
    out, err, rc = bench.run_group([sys.executable, "-c", "print('ok')"], 30)
    assert rc == 0 and out.strip() == "ok"


# ---- round 4: one seeded frame set over N ranks, cross-rank equality, per-rank PCIe-inclusive legs, CPU affinity ------------

def _worker_cross(rank, world, port, F, corrupt, out_q):
    """What bench.py does around its timed region, with the oracle standing in for each rank's GPU: frames of ONE seeded set
    dealt f mod world, 16-sample hashes merged, rank 0 re-filters two frames of every other rank, per-rank e2e figures gathered
    and turned into the whole-job rates."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import importlib.util
    import torch.distributed as dist
    from gpu_video_codec_amd import shard
    from oracle import oracle
    spec = importlib.util.spec_from_file_location("bench_x", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gidx = shard.global_frames_of_rank(F, rank, world)
    frames = bench.make_frames(64, 48, F * world, 8, seed=1, indices=gidx)
    local = {}
    for f in range(0, F, max(1, F // 16)):
        out = oracle.filter_plane(frames[f], 32)
        if corrupt and rank == 1 and f == 0:
            out = out.copy()
            out[5, 5] ^= 1      # one wrong sample on rank 1's "GPU"
        local[gidx[f]] = hashlib.sha256(out.tobytes()).hexdigest()

    def refilter(g):
        y = bench.make_frames(64, 48, F * world, 8, seed=1, indices=[g])[0]
        return hashlib.sha256(oracle.filter_plane(y, 32).tobytes()).hexdigest()
    cross = shard.frames_equal_across_ranks(dist, rank, world, local, refilter)
    per_rank = shard.gather_objects(dist, {"rank": rank, "e2e_sequence_frames": 24, "e2e_sequence_s": 0.010 * (rank + 1),
                                           "e2e_host_frame_s": 0.0004 * (rank + 1)})
    total = shard.aggregate_rate(per_rank, "e2e_sequence_frames", "e2e_sequence_s")
    dist.barrier()
    dist.destroy_process_group()
    out_q.put((rank, gidx, cross, total, frames[:2].tobytes()))


@pytest.mark.parametrize("corrupt", [False, True])
def test_two_ranks_share_one_frame_set_and_rank0_rechecks_the_others(corrupt):
    import importlib.util
    import torch.multiprocessing as mp
    spec = importlib.util.spec_from_file_location("bench_y", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    F, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90) + (50 if corrupt else 0)
    procs = [ctx.Process(target=_worker_cross, args=(r, world, port, F, corrupt, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = bench.make_frames(64, 48, F * world, 8, seed=1)       # what ONE rank of a 12-frame job holds
    for rank, gidx, cross, total, first_two in res:
        assert gidx == list(range(rank, F * world, world))
        assert first_two == whole[gidx[:2]].tobytes()               # the same frames, wherever they are filtered
        assert cross["frames_hashed"] == 2 * F and cross["frames_rechecked_on_rank0"] == 2
        assert cross["frames_equal_1gpu"] == (not corrupt)
        assert cross["mismatches"] == ([1] if corrupt else [])      # global frame 1 = rank 1's first frame
        assert total == pytest.approx(48 / 0.020)                   # all ranks' frames / the slowest rank's time


def test_gpu_cpu_topology_from_sysfs(tmp_path, monkeypatch):
    """shard.cpus_near_gpu / pin_to_gpu_cpus read the KFD topology and the PCI function's local_cpulist -- no HIP call, so a
    rank can pin itself before it initialises the runtime.  A fake sysfs tree: two CPU nodes, two GPUs on different sockets."""
    from gpu_video_codec_amd import shard
    nodes = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes"
    props = {0: "cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n",
             1: "cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n",
             2: "cpu_cores_count 0\nsimd_count 1024\nlocation_id 8960\ndomain 0\n",      # 0x2300 -> 0000:23:00.0
             3: "cpu_cores_count 0\nsimd_count 1024\nlocation_id 49408\ndomain 1\n"}     # 0xc100 -> 0001:c1:00.0
    for n, text in props.items():
        (nodes / str(n)).mkdir(parents=True)
        (nodes / str(n) / "properties").write_text(text)
    for pci, cpus in (("0000:23:00.0", "0-3,128-131\n"), ("0001:c1:00.0", "64-67\n")):
        d = tmp_path / "bus" / "pci" / "devices" / pci
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(cpus)
    root = str(tmp_path)
    assert shard.gpu_pci_ids(root) == ["0000:23:00.0", "0001:c1:00.0"]
    assert shard.parse_cpulist("0-2,7,9-10") == {0, 1, 2, 7, 9, 10} and shard.parse_cpulist("\n") == set()
    assert shard.cpus_near_gpu(0, root, env={}) == {0, 1, 2, 3, 128, 129, 130, 131}
    assert shard.cpus_near_gpu(1, root, env={}) == {64, 65, 66, 67}
    assert shard.cpus_near_gpu(0, root, env={"HIP_VISIBLE_DEVICES": "1"}) == {64, 65, 66, 67}    # device 0 of the process = GPU 1
    assert shard.cpus_near_gpu(2, root, env={}) == set()
    assert shard.cpus_near_gpu(0, root, env={"HIP_VISIBLE_DEVICES": "GPU-abc"}) == set()          # UUID lists are left alone
    assert shard.cpus_near_gpu(0, str(tmp_path / "nothing"), env={}) == set()
    # pinning: intersected with what the process may use; nothing to intersect -> nothing changed
    before = os.sched_getaffinity(0)
    try:
        applied = shard.pin_to_gpu_cpus(0, root)
        assert applied == sorted(before & {0, 1, 2, 3, 128, 129, 130, 131})
        assert os.sched_getaffinity(0) == (set(applied) if applied else before)
    finally:
        os.sched_setaffinity(0, before)
    assert shard.pin_to_gpu_cpus(0, str(tmp_path / "nothing")) == []
