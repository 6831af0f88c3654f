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


def test_two_rank_bench_line_is_judgeable_and_verifies_itself():
    """`bench.py --gpus 2` end to end on CPU: the SAME main() the MI355X run executes, with gloo ranks, the kernels behind the
    SIMT emulator and a tiny model (--selftest-emu).  The N > 1 line must still carry `roofline` (rank 0's own launches) and `cpu_baseline`, the
    parity verdict of the float64 arbiter, and the run must have verified itself: replicas bit-identical after the timed steps (CRC32 of theta /
    Adam m / Adam v / BatchNorm buffers, all-gathered), the all-reduced outer gradient equal to the single-handle 8-task gradient, and report the
    exchange (ranks, exposed ms, collectives per step)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-emu", "--steps", "1", "--warmup", "0", "--no-second-order"],
                         env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["tasks_per_gpu"] == 4 and "selftest" in d
    assert d["allreduce_ms_per_step"] is not None and d["allreduce_overlap"]["collectives_per_step"] >= 1 and d["allreduce_payload_mbytes"] > 0
    assert d["roofline"]["tasks_in_the_launches"] == 4 and d["roofline"]["launches"] > 0 and "all_gemm" in d["roofline"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["speedup_vs_cpu_baseline"] is not None
    assert d["replicas"]["bit_identical"] is True and d["replicas"]["ranks_compared"] == 2 and set(d["replicas"]["crc32"]) == {"theta", "adam_m", "adam_v", "bn_buffers"}
    ex = d["outer_gradient_vs_single_handle"]
    assert ex["ok"] is True and ex["max_rel"] <= 1e-5 and ex["synced_losses_rel"] <= 1e-5      # (emulator: same kernels, serial streams — roundoff only)
    pc = d["parity_check"]
    assert pc["grad_pass"] is True and pc["tasks_checked"] == 4 and pc["max_rel"] < 1e-4
    assert pc["grad_err_engine_vs_fp64"] is not None and pc["grad_err_oracle32_vs_fp64"] is not None and len(pc["per_task"]) == 4


def test_emulate_world_reports_the_emulated_share():
    """`--emulate-world 8` on one process holds ONE task in its launches and says so (config.tasks_per_gpu was 8 in round 5)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-emu", "--emulate-world", "8", "--steps", "1", "--warmup", "0",
                          "--no-second-order", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][0]
    assert d["n_gpus"] == 1 and d["emulated_world"] == 8 and d["config"]["tasks_per_gpu"] == 1 and d["roofline"]["tasks_in_the_launches"] == 1
    assert d["replicas"] is None and d["outer_gradient_vs_single_handle"] is None


def test_replica_check_bites():
    """The self-verification is not decorative: one float of one rank's weights off by 1e-6 after the timed steps and the run refuses to print a line."""
    env = _env()
    env["MTTS_SELFTEST_BREAK_REPLICA"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-emu", "--steps", "1", "--warmup", "0", "--no-second-order",
                          "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0
    assert "replicas diverged" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
