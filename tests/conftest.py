import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config._shard_job = None
    # tests/test_shard_gpu.py: its two worker ranks are child processes that use the GPU.  They are started HERE, before
    # this process has made any GPU call (a process that has initialised the GPU must not start them), and only when
    # the GPU tests are selected on a box that has a device node.
    markexpr = getattr(config.option, "markexpr", "") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr and os.path.exists("/dev/kfd"):
        import subprocess
        import tempfile

        tmp = tempfile.mkdtemp(prefix="mslam_shard_")
        out = os.path.join(tmp, "result.json")
        port = 29000 + os.getpid() % 2000
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs, logs = [], []
        for r in range(2):
            lg = os.path.join(tmp, f"rank{r}.log")
            logs.append(lg)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), "2",
                                           str(port), out], stdout=open(lg, "w"), stderr=subprocess.STDOUT, env=env))
        config._shard_job = (procs, out, logs)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import mslam_hip

    mslam_hip.check(mslam_hip.lib().mslam_device_check(), "device_check")
    return torch.device("cuda:0")
