"""`python bench.py --gpus N` must start its own ranks when no launcher did (the driver's N = 1 form of the command with
N > 1), and must run unchanged under torch.distributed.run.  CPU: the launch protocol only (`--launch-check`: spawn,
rendezvous on 127.0.0.1, barriers, MAX over ranks, ONE JSON line from rank 0).  GPU: the real two-rank step with two ranks
sharing the box's one GPU (gloo carries the collectives)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_plain_command_spawns_its_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-check"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["max_rank_seen"] == 1 and lines[0]["steps"] == 3


def test_under_torch_distributed_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--launch-check"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_world_size_mismatch_is_refused():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=120, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_end_to_end():
    # (reduced shapes: the test is about the launch protocol and the two-rank step, not about the benchmark configuration)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-kernel-timer", "--size", "224", "--batch", "4"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["final_loss"] == d["config"]["final_loss"]          # not NaN


@pytest.mark.gpu
def test_eight_ranks_on_one_gpu_launch_protocol():
    """The first real 8-GPU SCALE run must not die in set-up: the N = 8 form of the driver's command on the one GPU of the test box
    (gloo carries the host collectives, the eight ranks share the device; tiny model, batch 1, 64x64 - the launch protocol, the
    mailbox set-up with world 8, SyncBN exchanges and the gradient exchange over eight ranks, not the benchmark configuration)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--backend", "gloo", "--spec", "tiny", "--size", "64", "--batch", "1",
                        "--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--no-kernel-timer"], env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1200, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["final_loss"] == d["config"]["final_loss"]          # not NaN
    assert d["config"]["syncbn_peer_timeout"] in (False, None)


def test_one_rank_ddp_module_run_is_parsed_from_a_noisy_stdout(monkeypatch):
    """bench.module_path_subprocess: the one-rank DistributedDataParallel timings of the bench line run in processes of their own
    because RCCL prints a banner on stdout; the record is the first JSON line of that stdout, whatever surrounds it"""
    sys.path.insert(0, ROOT)
    import bench
    from types import SimpleNamespace as NS
    line = json.dumps({"ms_per_step": 21.5, "value": 372.0, "steps": 20,
                       "config": {"optimizer": "cris.pytorch_amd.optim.Adam (fused update: True)", "ddp_one_rank": True, "replay": "graph",
                                  "final_loss": 0.5}})
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env")
        return NS(returncode=0, stdout="RCCL version : 2.26.6\nHIP version  : 7.0\n" + line + "\nLibrccl path : /x/librccl.so\n", stderr="")
    monkeypatch.setattr(subprocess, "run", fake_run)
    args = NS(module_steps=20, batch=8, size=416, spec="r50", word_len=None)
    rec = bench.module_path_subprocess(args, "cris")
    assert rec["ms_per_step"] == 21.5 and rec["ddp_one_rank"] is True and rec["own_process"] is True
    assert "--ddp-one-rank" in seen["cmd"] and seen["cmd"][seen["cmd"].index("--optimizer") + 1] == "cris"
    assert not any(k in seen["env"] for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: NS(returncode=1, stdout="", stderr="boom"))
    assert "error" in bench.module_path_subprocess(args, "torch")
