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
