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


def test_single_process_multi_gpu_mode():
    """`python bench.py --gpus 2` without a launcher: both shards driven from one process through gf_multi_*_dev
    (GF_BENCH_SHARE_GPU test hook: both contexts on the box's one device)."""
    env = dict(os.environ, GF_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "dem1024",
           "--cpu-sample-tiles", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["bit_exact"] is True and d["value"] > 0
    assert "gf_multi" in d["config"]["processes"]


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)


def test_default_line_has_baselines_and_host_path():
    """the default bench line: roofline with replayed traffic provenance, CPU baseline on one core AND on every core
    (native threads), the host-memory path"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--workload", "dem1024",
                        "--cpu-sample-tiles", "128"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["bit_exact"] is True and d["n_gpus"] == 1
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0
    assert 1 <= cb["all_cores"]["cores"] <= os.cpu_count() and cb["all_cores"]["value"] > cb["value"] * 0.5
    assert d["host_path"]["roundtrip_GBps"] > 0
    assert "traffic_replayed_from" in d["roofline"]


@pytest.mark.parametrize("codec,workload", [("canon", "dem1024"), ("lsop", "dem1024"), ("float", "float256")])
def test_single_process_multi_gpu_mode_other_codecs(codec, workload):
    """The same for CodecCanonHuffman, LSOP12 and CodecFloat (BASELINE config 5): `--gpus 2` without a launcher."""
    env = dict(os.environ, GF_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload,
           "--codec", codec, "--cpu-sample-tiles", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["bit_exact"] is True and d["value"] > 0
    if codec != "float":
        assert len(d["devices"]) == 2
