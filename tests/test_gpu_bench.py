"""GPU tests of bench.py's multi-rank path: `--gpus N` starts its ranks itself, n_gpus is the number of ranks that joined the exchange,
and every launch's results travel to rank 0.  On a one-GPU box both ranks sit on device 0 and exchange through an IPC mapping of rank
0's buffers (RCCL refuses two ranks on one device); with two or more devices the same command goes over RCCL (ncclSend/ncclRecv)."""
import json
import subprocess
import sys

import pytest

import methyldackel_amd as mdk
from conftest import REPO

pytestmark = pytest.mark.gpu


def run_bench(*extra):
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--passes", "2", "--no-cpu-baseline"] + list(extra),
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_two_ranks_on_one_device_exchange_every_launch():
    j = run_bench("--devices", "0,0")
    assert j["n_gpus"] == 2
    assert j["exchange"]["exchanges"] == j["config"]["launches_per_step_per_gpu"] > 0
    assert "IPC" in j["exchange"]["transport"]
    assert j["value"] > 0 and j["scaling"] == "weak"


def test_two_devices_exchange_over_rccl():
    if mdk.lib_hip().md_dev_count() < 2:
        pytest.skip("needs two GPUs: ncclSend/ncclRecv between devices")
    j = run_bench()
    assert j["n_gpus"] == 2 and j["exchange"]["exchanges"] > 0
    assert "ncclSend" in j["exchange"]["transport"]


def test_two_ranks_without_the_exchange():
    """what the run falls back to when a communicator cannot be made: the ranks still do their interval-sharded work, and the line says so"""
    j = run_bench("--devices", "0,0", "--no-exchange")
    assert j["n_gpus"] == 2 and j["value"] > 0
    assert j["exchange"]["exchanges"] == 0 and j["exchange"]["transport"].startswith("NONE")


def test_the_command_as_two_ranks_leg(tmp_path):
    """bench.py's e2e_ranks leg (the thing that shards: `MethylDackel extract` as N ranks, csrc/host/mdk_ranks.c) on a small two-contig sample, both ranks
    on device 0: dealt and claimed, each identical to the oracle's output"""
    import argparse, time
    sys.path.insert(0, str(REPO))
    import bench
    args = argparse.Namespace(budget_s=600.0, coverage=30.0, synth_args="", large_sample_length=4_000_000, xl_copies=2)
    result = {}
    G = bench.Legs(args, result, tmp_path, tmp_path, [], time.time(), mdk)
    sp = G.synth(tmp_path / "two", "4000000,3000000", 30.0, 11, par=4)
    t_c, _, d_c, _ = G.run_oracle(sp, "cpu", 8, 250_000, 1)
    t1, _, d1, ok1, _ = G.run_ours(sp, "one", {}, runs=1)
    assert ok1 and G.same(d1, d_c)
    bench.ranks_on_sample(G, result, sp, 2, {"seconds": t1, "calls": bench._calls_of(d_c), "sample_bp": 7_000_000}, d_c)
    e = result["e2e_ranks"]
    assert e["ranks"] == 2 and e["dealt"]["ok"] and e["dealt"]["identical_to_oracle"] and e["claimed"]["ok"] and e["claimed"]["identical_to_oracle"]
    assert "TCP" in e["exchange"]["transport"] or "ncclSend" in e["exchange"]["transport"]
