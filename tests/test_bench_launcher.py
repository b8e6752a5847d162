"""bench.py --gpus N as its own launcher: a plain `python bench.py --gpus N` must not depend on
torch.distributed.run.  CPU tests drive the launcher, the rendezvous and the reduction with `--dry-run` (no kernel, no
GPU, `value` null -- plumbing only); the GPU test runs the real replica step on two ranks that share cuda:0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def _one_line(stdout: str):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_plain_process_spawns_its_ranks_and_prints_one_line(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == n and d["dry_run"] is True and d["value"] is None
    assert d["config"]["spawned_by_bench"] is True


def test_torchrun_path_still_works():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", BENCH, "--gpus", "2", "--dry-run",
                        "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["spawned_by_bench"] is False


def test_a_failing_rank_fails_the_job_without_a_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--dry-fail-rank", "1", "--launch-timeout", "60"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert r.stdout.strip() == ""


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--no-loop"], capture_output=True, text=True, timeout=300,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_two_replica_ranks_on_one_gpu_from_a_plain_process():
    """The real step (asd_verify_accept_fused on resident logits) on two self-spawned ranks sharing cuda:0; control over gloo."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--steps", "20", "--warmup", "5",
                        "--no-loop", "--no-cpu-baseline", "--no-other-workloads"], capture_output=True, text=True,
                       timeout=600, env=_env(ASD_BENCH_ONE_DEVICE="1"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] < 1.0


@pytest.mark.parametrize("n,kind", [(2, "tiers"), (4, "tiers"), (8, "sharded-target")])
def test_default_multi_gpu_line_carries_the_loop_sub_record_of_its_configuration(n, kind):
    """the BARE command (`bench.py --gpus N`, what the driver runs) must exercise BASELINE configs[3] at
    N = 2 / 4 (tiers placed over the ranks, the reference's configs/qwen3_models.yaml:5-53) and configs[4] at N = 8 (replicated
    drafts + sharded target).  The dry run walks the same branch and prints the sub-record's keys with null values."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    loop = d["loop"]
    assert loop["kind"] == kind and loop["rccl_ranks"] == n
    for key in ("kind", "placement", "rccl_ranks", "backend", "verified_tokens_per_s", "ms_per_step", "steps", "roofline", "bytes_sent"):
        assert key in loop, key
    if kind == "tiers":
        want = {2: [[0], [1]], 4: [[1], [2, 3]]}[n]
        assert loop["placement"]["tiers"] == want and loop["placement"]["draft"] == 0
    else:
        assert "sharded over 8 ranks" in loop["placement"] and "batch 128" in loop["placement"]
    assert d["sharded_verify"]["ranks"] == n


def test_a_rank_that_hangs_in_the_data_path_fails_the_job_but_the_headline_is_printed():
    """A collective that never completes: the watchdog prints the headline with the error recorded in the sub-record that was
    running and EVERY rank exits non-zero (round 3 left with exit code 0)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--dry-stall-rank", "1", "--multi-gpu-timeout", "4",
                        "--launch-timeout", "90"], capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and "timed out" in d["loop"]["error"]


def test_a_rank_whose_sub_record_raises_ends_the_job_in_seconds_not_at_the_time_bound():
    """One rank raises in the data-path phase while the other waits in the exchange: the report goes through the rendezvous store,
    the headline is printed with that rank's error and every rank exits non-zero -- long before --multi-gpu-timeout."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--dry-raise-rank", "1", "--multi-gpu-timeout", "120",
                        "--launch-timeout", "200"], capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert time.monotonic() - t0 < 60
    d = _one_line(r.stdout)
    assert "rank 1" in d["loop"]["error"] and "injected" in d["loop"]["error"]


def test_a_sub_record_that_raises_on_every_rank_is_recorded_and_the_job_ends_normally():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--dry-raise-rank", "2", "--multi-gpu-timeout", "120",
                        "--launch-timeout", "200"], capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert "injected" in d["loop"]["error"] and d["loop"]["kind"] == "tiers"
