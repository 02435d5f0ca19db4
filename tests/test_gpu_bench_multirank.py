"""bench.py's multi-rank control flow (rank-sharded tile ranges, timing barrier, max-over-ranks, rank-0 JSON line) on
the one GPU of the test box: two ranks share device 0 over gloo (GF_BENCH_BACKEND test hook; RCCL refuses two ranks on
one device -- the judged multi-GPU runs use RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_json_line():
    env = dict(os.environ, GF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "dem1024", "--cpu-sample-tiles", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["bit_exact"] is True
    assert d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0
