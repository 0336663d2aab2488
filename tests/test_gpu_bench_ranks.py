"""bench.py with more than one rank on the one-GPU box: its N > 1 control flow (rendezvous, ray sharding, per-rank scene build,
the chunked double-buffered trace -> gather pipeline, barrier + max-over-ranks timing, rank 0's single JSON line) under
torch.distributed.run exactly as the driver launches it -- with `--backend gloo`, bench.py's test mode in which every rank uses
device 0 and the hit records travel through torch.distributed.gather over host memory.  The native RCCL gather of the real
N > 1 run (vt_gather_hits_dev) is exercised with one rank in test_native_gather_single_rank and with 2 - 8 ranks against the test
double in tests/test_gpu_fake_group.py; what only a multi-GPU node can add are RCCL's kernels and the links."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nranks, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--no-cpu", "--no-pmc", "--alt-builder", "none", "--legs", "off"] + extra
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    # ... and nothing else reaches stdout: the ranks send everything but the result (RCCL's banner comes from C code) to stderr
    assert [ln for ln in p.stdout.splitlines() if ln.strip()] == lines, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("nranks", [2, 3])
def test_weak_scaling_line_with_several_ranks(nranks):
    d = _run(nranks, ["--side", "512"])
    assert d["n_gpus"] == nranks and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["rays_per_gpu"] == 512 * 512 and d["config"]["rays_total"] == nranks * 512 * 512
    assert d["value"] > 0 and abs(d["value"] - d["config"]["rays_total"] / d["ms_per_step"] / 1e3) <= 1e-2 * d["value"]
    assert d["vs_baseline"] is None and d["unit"] == "Mrays/s" and "gather" in json.dumps(d["config"])


def test_strong_scaling_line_with_two_ranks():
    """configs[4]'s shape (camera tiles split over the ranks) on a small scene."""
    d = _run(2, ["--scaling", "strong", "--scene", "S100k", "--tiles", "4"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rays_total"] == 4 * 1024 * 1024 and d["value"] > 0


def test_bare_command_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` exactly as the driver runs N = 1 -- no launcher: bench.py starts the two ranks as a child
    process before it touches HIP and relays rank 0's line (the only thing on stdout)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--no-cpu", "--no-pmc", "--alt-builder", "none", "--legs", "off", "--side", "512"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out) == 1, "stdout carries the result line and nothing else"
    d = json.loads(out[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["rays_total"] == 2 * 512 * 512 and d["value"] > 0


def test_bare_command_reports_a_crashed_rank():
    """A rank that dies (here: an unknown scene name) is a non-zero exit code and a one-line reason, never a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--no-cpu", "--no-pmc",
                        "--scene", "no_such_scene"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert [ln for ln in p.stderr.splitlines() if ln.strip()][-1].startswith("[bench] FATAL: the ranks exited with code")


def test_one_rank_runs_the_native_rccl_control_flow():
    """--force-dist: the N > 1 control flow of the per-rank form with REAL RCCL and one rank (communicator, vt_gather_hits_dev on the
    communication stream, double-buffered pipeline, reserved CUs, the one-batch-in-pieces measurement behind its watchdog): stdout
    carries the result line and nothing else (RCCL's banner goes to stderr), the gather is verified."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "5", "--warmup", "2", "--no-cpu", "--no-pmc",
                        "--alt-builder", "none", "--side", "1024"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and out[0].startswith("{"), p.stdout[-2000:]
    d = json.loads(out[0])
    bd = d["config"]["dist_breakdown"]
    assert d["config"]["gather_verified"] is True and "native ncclGather" in bd["gather_kind"]
    assert set(bd["single_batch_ms"]) == {"1", "2", "4", "8"} and d["config"]["launch_options"]["reserved_cus"] == 32


def test_reserved_cu_probe_runs_with_one_rank():
    """--reserve-cus probe: the auto-tuning step of an N > 1 run (64 / 32 / 16 / 0 reserved CUs timed with the gather in flight, the
    fastest kept) executed with one rank -- there is no transfer to hide here, so it must settle on a small reservation, and the
    line must say what it measured."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--reserve-cus", "probe", "--steps", "5", "--warmup", "2", "--no-cpu",
                        "--no-pmc", "--alt-builder", "none", "--side", "2048"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    probe = d["config"]["dist_breakdown"]["reserved_cus_probe"]
    assert set(probe["ms_per_step"]) == {"64", "32", "16", "0"} and all(v > 0 for v in probe["ms_per_step"].values())
    assert probe["chosen"] in (0, 16, 32) and d["config"]["launch_options"]["reserved_cus"] == probe["chosen"]
    assert probe["ms_per_step"]["64"] >= probe["ms_per_step"]["0"] * 0.98        # reserving CUs never makes the lone trace faster
    assert d["config"]["gather_verified"] is True
