"""bench.py --gpus N starts its N ranks itself (no launcher), refuses a world size that differs from N, and fails
loudly without a GPU.  The rank logic is exercised here over gloo with a stand-in child script; the real thing
(`GMR1_BENCH_BACKEND=gloo python bench.py --gpus 2` on one MI355X) is the GPU test at the bottom."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_spawn_ranks_starts_n_children(tmp_path):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent("""
        import json, os, sys
        import torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
        dist.init_process_group("gloo", rank=rank, world_size=world)      # env:// rendezvous, as bench.py does
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t), "argv": sys.argv[1:]}), flush=True)
        dist.destroy_process_group()
    """))
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(3, argv=['--gpus', '3'], script=%r, timeout=150))" % (ROOT, str(child)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out == {"n_gpus": 3, "sum": 6.0, "argv": ["--gpus", "3"]}


def test_spawn_ranks_reports_a_failed_rank(tmp_path):
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(2, argv=[], script=%r, timeout=100))" % (ROOT, str(child)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 7                               # and it did not wait for the sleeping rank


def test_bench_refuses_wrong_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-cpu"], capture_output=True, text=True, timeout=120,
                       env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_gpus2_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("this box has a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-cpu"], capture_output=True, text=True, timeout=300,
                       env=env)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr                       # no CPU fallback, no line with n_gpus: 1
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gather_bytes_and_block_partition_world2(tmp_path):
    """What `bench.py --workload nt3|tch3 --gpus N` does around its kernels, over gloo with two ranks: every rank takes
    shard.partition_contiguous(n, N, r, group=4) of one global list (block edges on FACCH3 groups) and rank 0 gets the
    ranks' result bytes back concatenated in rank order (bench.gather_bytes), ragged lengths included."""
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent("""
        import json, os, sys
        import numpy as np
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        from __graft_entry__ import load_package
        pkg = load_package()
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        n = 1000 * 40 + 36                                   # a ragged global count
        g0, g1 = pkg.shard.partition_contiguous(n, world, rank, group=4)
        assert g0 %% 4 == 0 and (g1 %% 4 == 0 or g1 == n)
        glob = (np.arange(n, dtype=np.uint32) * 2654435761).astype(np.uint32)      # stands for per-burst results
        mine = glob[g0:g1]
        got = bench.gather_bytes(mine, rank, world, "gloo", torch.device("cpu"))
        if rank == 0:
            print(json.dumps({"same": bool(np.array_equal(got.view(np.uint32), glob)), "blocks": [g0, g1]}), flush=True)
        else:
            assert got is None
        dist.destroy_process_group()
    """ % ROOT))
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(2, argv=[], script=%r, timeout=150))" % (ROOT, str(child)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["same"] and out["blocks"] == [0, 20020]


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("workload", ["nt3", "tch3"])
def test_bench_configs4_gpus2_over_gloo_on_one_gpu(workload):
    """`GMR1_BENCH_BACKEND=gloo python bench.py --workload nt3|tch3 --gpus 2`: BASELINE configs[4] on more than one rank --
    one global workload in contiguous blocks, the line says n_gpus 2 / scaling strong, and rank 0 found the ranks'
    concatenated speech frames and FACCH3 results equal to ONE run over the whole workload."""
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: not starting child processes from it")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["GMR1_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, BENCH, "--workload", workload, "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--bursts", "1204", "--preroll-s", "0", "--no-cpu"],
                       capture_output=True, text=True, timeout=550, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["unit"] == "Mbursts/s"
    assert out["checks"]["sharded_outputs_identical_to_single_gpu_run"] is True
    if workload == "nt3":
        assert out["config"]["global_bursts"] == 12040 and out["config"]["bursts_on_rank_0"] == 6020
        assert out["checks"]["facch3_payloads_match_sent"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_gpus2_over_gloo_on_one_gpu():
    """`GMR1_BENCH_BACKEND=gloo python bench.py --gpus 2`: two ranks share the one MI355X, the exchanges run over
    gloo.  One JSON line with n_gpus 2, and the sharded config-4 keys (scatter, receive loop, gather; carrier 0's
    frames identical to the oracle's loop).  Child processes are started only while this process has not brought
    up the GPU (this file sorts first in the suite)."""
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: not starting child processes from it")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["GMR1_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--bursts", "7000",
                        "--preroll-s", "0", "--shard-arfcns", "5", "--shard-seconds", "4"],
                       capture_output=True, text=True, timeout=550, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_bursts"] == 14000
    assert out["checks"]["payloads_match_sent"]
    sh = out["sharded_rx"]
    assert sh["ranks_seen"] == 2 and sh["backend"] == "gloo"
    assert sh["frames"] > 0 and sh["carriers_with_frames"] == 5
    assert sh["frames_identical_to_oracle"] and sh["tiles_of_carrier0_identical_across_ranks"]
    for k in ("scatter_ms", "rx_loop_ms", "gather_ms"):
        assert sh[k] > 0
    # every record of every carrier: the sharded exchange returns what ONE gmr1_hip_rx_run over all carriers returns
    assert sh["records_identical_to_single_gpu_run"] and sh["single_gpu_run_frames"] == sh["frames"]
    # and the other way to feed the ranks (each holds its own carriers, no scatter) returns the same records
    assert sh["resident"]["records_identical_to_scattered_run"] and sh["resident"]["rx_loop_ms"] > 0
    lo, hi = sh["rx_loop_ms_per_rank_min_max"]
    assert 0 < lo <= hi


def test_side_runner_protocol():
    """`bench.py --side-runner` is the helper the default run starts before it touches the GPU: it waits for "go" on stdin and
    leaves without doing anything if its parent goes away first (stdin closes) -- it must never touch the GPU by itself."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--side-runner"], input="", capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == ""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def side_runner_main"):src.index("def side_workloads")]
    assert "torch" not in body and "api." not in body
