"""`python bench.py --gpus N` is its own launcher (the driver runs exactly that command): the parent starts N ranks
of itself before anything touches a GPU, relays rank 0's single JSON line and exits with the failing rank's code.
Exercised here with gloo ranks on a stub step (`--launch-selftest`, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra_env=None, n=2):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--launch-selftest", "--steps", "4",
                           "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_launcher_starts_n_ranks_and_relays_one_line():
    r = _run()
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # rank chatter on stdout does not reach the result stream
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 4 and d["warmup"] == 1
    assert len(d["ms_per_step_by_rank"]) == 2
    # the slowest rank's time is the one that counts
    assert abs(d["ms_per_step"] - max(d["ms_per_step_by_rank"])) < 1e-9
    assert b"launcher: 2 ranks started" in r.stderr


def test_launcher_three_ranks():
    r = _run(n=3)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert json.loads(r.stdout.decode())["ranks_seen"] == 3


def test_a_dying_rank_ends_the_run_with_its_code():
    """A rank that exits before the rendezvous: the others are stopped (not left waiting in a collective),
    nothing is printed on stdout, the return code is the failing rank's."""
    r = _run({"MTAM_SELFTEST_FAIL_RANK": "1"})
    assert r.returncode == 3, (r.returncode, r.stderr.decode()[-2000:])
    assert r.stdout.decode().strip() == ""


def test_parent_never_touches_the_gpu_stack():
    """The launcher branch runs before any torch / package import: a parent that initialised HIP could not start
    workers on the GPU pool.  Checked on the source: nothing between the argument parse and launch() imports."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    head = main[:main.index("sys.exit(launch(")]
    assert "import torch" not in head and "mtamrecommender_amd" not in head and "__graft_entry__" not in head
    body = src[src.index("def launch("):src.index("def selftest_rank(")]
    assert "import torch" not in body and "mtamrecommender_amd" not in body


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-selftest"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"WORLD_SIZE=3" in r.stderr


def test_under_torch_distributed_run_the_file_is_one_rank():
    """The driver may also start it as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: WORLD_SIZE
    is then set, no launcher runs, every process is one rank and rank 0 prints the line."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29713", BENCH, "--gpus", "2",
                        "--launch-selftest", "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2
    assert b"launcher:" not in r.stderr
