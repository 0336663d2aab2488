"""`python3 bench.py --gpus N` without a launcher starts its N ranks itself, as a child process (bench.py: launch_ranks).
Here, without a GPU, the ranks refuse to run ("bench.py needs a HIP device"): what can be checked on the CPU is that the parent
never falls back to anything -- non-zero exit code, nothing on stdout, a one-line reason on stderr -- and that it gets there
without importing torch itself (the ranks must be a child of a process that has not touched HIP).  The successful case (two
ranks on the one-GPU box, one well-formed line) is tests/test_gpu_bench_ranks.py::test_bare_command_starts_its_own_ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_multi_gpu_command_fails_loudly_without_a_device():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--no-cpu", "--no-pmc"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == "", "no result line may appear when the ranks failed"
    last = [ln for ln in p.stderr.splitlines() if ln.strip()][-1]
    assert last.startswith("[bench] FATAL: the ranks exited with code") and "2 ranks" in last
    assert "starting 2 ranks as a child process" in p.stderr


def test_launcher_runs_before_torch_is_imported():
    """The self-launch sits in front of `import torch` in main(): the parent of the ranks never initialises a GPU."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main()"):]
    assert body.index("sys.exit(launch_ranks(args))") < body.index("import torch")
    head = src[: src.index("def main()")]
    assert "\nimport torch" not in head and "\nfrom torch" not in head
    assert "os.exec" not in src, "never replace a process image: the ranks are a child process"
