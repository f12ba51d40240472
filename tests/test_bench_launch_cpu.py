"""bench.py refuses a rank count that does not match --gpus (it used to warn and run ONE rank).  Decided before any GPU
call, so this runs without a device."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_count_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr and p.stdout.strip() == ""


_GUARD = r"""
import json, sys, time
sys.path.insert(0, {root!r})
import bench
bench.barrier = lambda world: None                       # no device, no process group in this test
def fake(args, rank, world, dev, ranks_seen):
    {body}
bench.measure_sharded_backend = fake
out = {{"metric": "m", "value": 123.0}}
sb = bench.guarded_sharded_backend(None, 0, 2, None, 2, out, {replicas_done})
out["sharded_backend"] = sb
print(json.dumps(out))
"""


def _guard(body, replicas_done=True, timeout="0.5"):
    code = _GUARD.format(root=ROOT, body=body, replicas_done=replicas_done)
    env = dict(os.environ, BENCH_SHARD_TIMEOUT=timeout)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_a_failing_sharded_session_leaves_the_replica_line():
    """bench.py measures the replicas first and the sharded session behind a watchdog: when that session raises or does not
    finish, the ONE JSON line still carries the replica value (exit code 0); when it was the requested measurement
    (--mode shard-backend), the exit code says so."""
    import json

    p = _guard("return {'value': 7.0}")
    assert p.returncode == 0 and json.loads(p.stdout)["sharded_backend"] == {"value": 7.0}
    p = _guard("raise RuntimeError('boom')")
    d = json.loads(p.stdout)
    assert p.returncode == 0 and d["value"] == 123.0 and "RuntimeError: boom" in d["sharded_backend"]["error"]
    p = _guard("time.sleep(30)")
    d = json.loads(p.stdout)
    assert p.returncode == 0 and d["value"] == 123.0 and "did not finish" in d["sharded_backend"]["error"]
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1
    p = _guard("raise RuntimeError('boom')", replicas_done=False)
    assert p.returncode == 3
