"""`bench.py --gpus N` must really run N ranks: started as a plain process it launches them itself, started under
torch.distributed.run it checks WORLD_SIZE against --gpus.  CPU test of that plumbing (gloo, no filter: --dry-run)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out  # ONE line, from rank 0
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def test_plain_invocation_launches_the_ranks_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["dry_run"] is True and line["value"] is None


def test_under_torchrun_world_size_must_match_gpus():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), BENCH, "--dry-run"]
    ok = subprocess.run(base + ["--gpus", "2"], env=_env(), capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    assert _last_json(ok.stdout)["n_gpus"] == 2
    bad = subprocess.run(base + ["--gpus", "4"], env=_env(), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr


def test_single_rank_dry_run_reports_one_gpu():
    r = subprocess.run([sys.executable, BENCH, "--dry-run"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 1
