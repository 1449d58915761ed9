"""CPU: `python bench.py --gpus N` with no launcher in the environment starts N ranks itself (torch.distributed.run on 127.0.0.1,
one process per GPU) -- VERDICT round 3, "the flag is a trap".  The ranks here only join a gloo group (--launch-check): no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "KBNER_BENCH_CHILD")}
    return env


def test_gpus_flag_builds_the_torchrun_command():
    env = _env()
    env["KBNER_BENCH_LAUNCH_DRYRUN"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"     # torchrun picks the port itself
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "3", "--warmup", "1"]


def test_gpus_flag_without_launcher_runs_n_ranks_over_gloo():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 alone prints
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "n_gpus": 2, "world": 2, "ranks_seen": 2}


def test_with_world_size_in_the_environment_it_does_not_relaunch():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout.strip().splitlines()[-1])["world"] == 1


def test_newest_traffic_profile_matches_the_default_workload():
    """bench.py fills roofline.traffic from the newest profiles/*_hbm_traffic.json -- but only if that file was profiled at the
    micro-batch the default run uses (the file says so); a default change without a new profile set would silently print null"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kbner_bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pdir = os.path.join(ROOT, "profiles")
    newest = sorted(f for f in os.listdir(pdir) if f.endswith("_hbm_traffic.json"))[-1]
    tj = json.load(open(os.path.join(pdir, newest)))
    assert int(tj.get("micro_batch", 128)) == mod.DEFAULT_MICRO_BATCH, (newest, tj.get("micro_batch"))
    ks = [v for k, v in tj["kernels"].items() if "gemm256f_kernel" in k or "gemm256_kernel" in k]
    assert len(ks) == 3 and all(v["launches"] > 0 for v in ks), newest
