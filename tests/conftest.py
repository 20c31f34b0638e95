"""Shared test plumbing.  GPU tests are marked `gpu`; everything else runs on a CPU-only box."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
GOLDEN = REPO / "tests" / "golden"
ORACLE = REPO / "oracle" / "_build" / "mdk_oracle"
SYNTH = REPO / "tools" / "_build" / "mdk_synth"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The hot path first: under `-x` a failure in a side command (mbias, perRead, bed masks) must not keep the parity matrix of the
# `extract` path -- BASELINE.json's configs -- from running at all (round 4's driver run stopped at test 91 of 281 and never reached it).
GPU_ORDER = ["test_gpu_parity", "test_gpu_prep", "test_gpu_scaled_configs", "test_gpu_multi", "test_zoo", "test_gpu_edge_cases", "test_gpu_inflate",
             "test_gpu_bed", "test_gpu_perread", "test_gpu_mbias", "test_gpu_bench", "test_gpu_stress"]


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        mod = Path(str(it.fspath)).stem
        return (GPU_ORDER.index(mod) if mod in GPU_ORDER else len(GPU_ORDER), )
    items.sort(key=key)          # stable: the order inside a module, and of the CPU modules among themselves, is unchanged


@pytest.fixture(scope="session", autouse=True)
def built():
    """Everything is built in-tree by `make` (the GPU box receives the prebuilt .so files; make is then a no-op)."""
    r = subprocess.run(["make", "-C", str(REPO), "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return True


def run_oracle(args, cwd, dump=None):
    env = dict(os.environ)
    if dump:
        env["MDK_ORACLE_DUMP"] = str(dump)
    return subprocess.run([str(ORACLE), "extract"] + [str(a) for a in args], cwd=cwd, env=env, capture_output=True, text=True)


def synth(prefix, *args):
    r = subprocess.run([str(SYNTH), "-o", str(prefix)] + [str(a) for a in args], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout


def read_dump(path):
    """MDK_ORACLE_DUMP rows -> {(tid,pos): (type,isG,nmeth,nunmeth,noff,nvar)}"""
    d = {}
    for line in open(path):
        t = [int(x) for x in line.split()]
        d[(t[0], t[1])] = tuple(t[2:])
    return d


@pytest.fixture(scope="session")
def small_synth(tmp_path_factory):
    """A 60 kb, 2-contig, 25x noisy paired-end sample + a Bismark-style one, shared by several tests."""
    d = tmp_path_factory.mktemp("synth")
    synth(d / "pe", "-L", "40000,20000", "-c", "25", "-s", "11", "--extras", "--bbm", "--bw")
    synth(d / "bis", "-L", "30000", "-c", "20", "-s", "12", "--bismark")
    synth(d / "se", "-L", "20000", "-c", "15", "-s", "13", "--single")
    return d
