"""The N > 1 launch path: `python bench.py --gpus N` / `python run_task.py cfg.yml --gpus N` started WITHOUT a launcher (as the driver
starts bench.py) must bring up N ranks themselves (dp.self_launch: torch.distributed.run on 127.0.0.1, a free port) before anything touches
a GPU, pass rank 0's output through and return the ranks' exit code.  SURVEY 7 step 6 ("launcher, 1 proc / GPU"), 8e.  The CPU tests
use gloo and a rendezvous-only mode of bench.py; the GPU test runs the real benchmark with two ranks on the one GPU (gloo)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(VLTF_BENCH_RENDEZVOUS_ONLY="1", VLTF_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE line, rank 0's
    d = json.loads(lines[0])
    assert d == {"rendezvous_only": True, "world": 2, "gpus": 2, "rank_sum": 1.0}


def test_bench_under_an_external_launcher_does_not_launch_again():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=_env(VLTF_BENCH_RENDEZVOUS_ONLY="1", VLTF_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["world"] == 2


def test_self_launch_returns_the_ranks_exit_code(tmp_path):
    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        from vltf_amd import dp
        rc = dp.self_launch(int(sys.argv[1]))
        if rc is not None:
            sys.exit(rc)
        rank, world, _ = dp.init_from_env(backend="gloo")
        open(os.path.join(%r, "rank%%d_of_%%d" %% (rank, world)), "w").close()
        sys.exit(3 if (len(sys.argv) > 2 and rank == 1) else 0)
    """ % (ROOT, str(tmp_path))))
    r = subprocess.run([sys.executable, str(script), "3"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("rank") and "_of_" in f) == ["rank0_of_3", "rank1_of_3", "rank2_of_3"]
    r = subprocess.run([sys.executable, str(script), "2", "fail"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0                               # a failing rank fails the whole launch
    r = subprocess.run([sys.executable, str(script), "1"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rank0_of_1" in os.listdir(tmp_path)      # N = 1: no launcher, the process itself is rank 0


def test_the_process_sent_ahead_touches_this_ranks_card_only(monkeypatch):
    """N ranks under a launcher must not put N extra processes on card 0: each sends its process to its own card; without a
    launcher one process visits cards 0 .. gpus-1 in turn, and the ranks started afterwards inherit the marker."""
    from vltf_amd import dp
    seen = []
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: seen.append(cmd[-1]))
    monkeypatch.delenv("VLTF_GPU_PRETOUCHED", raising=False)
    monkeypatch.delenv("VLTF_NO_PRETOUCH", raising=False)
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert dp.pretouch_gpu(8) is True
    assert "for d in [3]:" in seen[-1]
    assert dp.pretouch_gpu(8) is False and len(seen) == 1              # the marker is set: not repeated
    monkeypatch.delenv("VLTF_GPU_PRETOUCHED")
    monkeypatch.delenv("LOCAL_RANK")
    assert dp.pretouch_gpu(4) is True
    assert "for d in [0, 1, 2, 3]:" in seen[-1]
    compile(seen[-1], "<pretouch>", "exec")
    monkeypatch.delenv("VLTF_GPU_PRETOUCHED")


@pytest.mark.gpu
def test_bench_gpus_2_runs_the_data_parallel_benchmark_by_itself():
    """The driver's command for N = 2, on the one GPU of the test box: gloo instead of RCCL (two ranks cannot share a GPU under RCCL),
    2 timed steps of the strong-scaling job (64 clips over 2 ranks)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-side", "--no-split-math"], env=_env(VLTF_DIST_BACKEND="gloo", VLTF_WGRAD_STREAM="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["dist_world_size"] == 2 and d["config"]["dist_backend"] == "gloo"
    assert d["config"]["grad_exchange"]["chunks_per_step"] == 7
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 64 and d["config"]["clips_per_gpu"] == 32
    assert d["value"] > 0 and d["steps"] == 2
