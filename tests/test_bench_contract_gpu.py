"""bench.py's output contract on a short run (a child process, as the driver starts it): ONE JSON line on stdout with the
keys the driver reads, the roofline object measured live (launches of the dominant shape were timed inside the region),
the per-kernel HBM entries, and the cpu_baseline object when it is not switched off."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line(device):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "4", "--preroll", "40",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"].startswith("SLAM frames/sec") and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 4 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6          # frames/s = 1 / (s per step) at N = 1
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert r["launches_timed"] > 0 and r["us_per_launch_avg"] >= r["us_per_launch_min"] > 0      # measured inside the timed region
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["gflop_per_launch"] * 1e3 / r["us_per_launch_avg"]) < 1e-6 * r["achieved"]
    assert {h["kernel"].split(" ")[0] for h in r["hbm_bound_kernels"]} >= {"prep_iter_proj_kernel", "iter_proj_kernel",
                                                                           "refine_matches_kernel<24>"}
    st = d["config"]["stats"]
    assert st["relocalised"] == 0 and st["keyframes"] >= 2 and st["encoder_rows"] >= 8
    assert "cpu_baseline" not in d                                                              # switched off above


def _check_two_rank_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]     # the libraries' banners went to stderr
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(device):
    """`python bench.py --gpus 2` as the driver starts it, WITHOUT a launcher: bench.py spawns one process per rank itself
    (before touching the GPU); rehearsed on one card (--share-gpu: both ranks on cuda:0, gloo collectives).  Default mode
    at N > 1 = both measurements: `value` is the whole-job aggregate of two replica sessions (both sessions' frames over
    the slowest rank's time, weak scaling) and `sharded_backend` is ONE session whose backend (pair inference +
    matching, GN with one all-reduce per iteration, TSDF voxels) is sharded over the two ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--preroll", "40",
           "--share-gpu"]
    d = _check_two_rank_line(subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT))
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 8 / 1e3)) < 1e-6 * d["value"]      # two sessions' frames / max time
    assert "2 rank(s) reported by the collective library" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d
    sb = d["sharded_backend"]
    assert sb["ranks"] == 2 and sb["ranks_reported_by_collective_library"] == 2 and sb["scaling"] == "strong"
    assert sb["value"] > 0 and abs(sb["value"] * sb["ms_per_step"] / 1e3 - 1.0) < 1e-6       # one session: frames/s = 1/(s per step)
    assert sb["keyframes"][1] >= 2 and sb["undirected_edges"][1] >= 1 and sb["relocalised"] == 0
    assert sb["allreduce_bytes_per_gn_iteration"] == 840 * 2 * sb["undirected_edges"][1]
    assert sb["broadcast_bytes_total"] > 0 and len(sb["announcements"]) >= 4


def test_bench_sharded_backend_under_the_launcher(device):
    """The same under torch.distributed.run (how the driver launches N > 1), sharded-backend measurement only."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4",
           "--preroll", "24", "--share-gpu", "--mode", "shard-backend"]
    d = _check_two_rank_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] == d["sharded_backend"]["value"]
