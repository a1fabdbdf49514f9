"""`python bench.py --gpus N` without a launcher around it: the parent starts the N ranks itself
(shard.launch_ranks).  Driven here with stub children -- no GPU, no torch in the children."""
import io
import json
import os
import subprocess
import sys
import time

from visual_odometry_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r"""
import json, os, sys, time
rank = int(os.environ["RANK"])
mode = sys.argv[1]
if mode == "env":
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
    if rank == 0:
        print(json.dumps({k: os.environ[k] for k in keys} | {"pad": "x" * 200000}))
    else:
        print("rank %d says something on stdout" % rank)      # must not reach the job's stdout
elif mode == "fail":
    if rank == 1:
        sys.exit(3)
    print("partial line from rank 0")
    sys.stdout.flush()
    time.sleep(120)                                            # the launcher has to end this rank
"""


def _stub(tmp_path):
    p = tmp_path / "stub_rank.py"
    p.write_text(STUB)
    return str(p)


def test_rank_env():
    e = shard.rank_env(3, 8, 29512, base={"KEEP": "1", "RANK": "99"})
    assert e["RANK"] == "3" and e["LOCAL_RANK"] == "3" and e["WORLD_SIZE"] == "8"
    assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29512" and e["KEEP"] == "1"


def test_launcher_relays_rank0_only(tmp_path):
    out = io.StringIO()
    rc = shard.launch_ranks(3, [sys.executable, _stub(tmp_path), "env"], out=out)
    assert rc == 0
    lines = out.getvalue().splitlines()
    assert len(lines) == 1                                     # one line, a long one (beyond a pipe buffer)
    got = json.loads(lines[0])
    assert got["RANK"] == "0" and got["LOCAL_RANK"] == "0" and got["WORLD_SIZE"] == "3"
    assert got["MASTER_ADDR"] == "127.0.0.1" and 1024 < int(got["MASTER_PORT"]) < 65536


def test_launcher_propagates_failure_and_stops_the_rest(tmp_path):
    out = io.StringIO()
    t0 = time.time()
    rc = shard.launch_ranks(2, [sys.executable, _stub(tmp_path), "fail"], out=out)
    assert rc == 3
    assert time.time() - t0 < 60                               # rank 0 did not sleep its two minutes
    assert out.getvalue() == ""                                # a failed job prints no result line


def test_bench_parent_starts_ranks_and_fails_with_them():
    """The real parent path of bench.py on a box without a GPU: both ranks fail at their first GPU call, the parent
    exits non-zero and prints no result line (and never asks for torch.distributed.run)."""
    if os.path.exists("/dev/kfd"):
        return                                                 # (on a GPU box this would be a real two-rank run)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "torch.distributed.run --nproc-per-node N for" not in p.stderr
