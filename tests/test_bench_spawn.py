"""bench.py --gpus N launch plumbing on CPU: started as ONE process it must spawn N ranks itself (the
reference's launcher, pl.Trainer(strategy="ddp"), main.py:30-38, does) and must never print an n_gpus it did not run on."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_spawn_command_shape():
    import bench
    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "2"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"]


def test_gpus_2_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher: two gloo ranks rendezvous on 127.0.0.1 and rank 0 reports n_gpus = 2."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-spawn"], env=_env(), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # only rank 0 prints
    assert lines[0]["n_gpus"] == 2 and lines[0]["allreduce_sum"] == 3.0 and lines[0]["tasks_per_rank"] == 4


def test_world_size_mismatch_is_refused():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")  # a launcher that gave us 1 rank while --gpus says 2
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-spawn"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "n_gpus" not in out.stdout
    # and the real (non-selftest) path refuses as well, before touching any GPU
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "n_gpus" not in out.stdout
