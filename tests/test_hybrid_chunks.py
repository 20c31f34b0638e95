"""CPU: the chunks `extract` hands out when pieces of the BAM are inflated "on the device" must be the chunks it hands out when the host
inflates everything -- same schedule, same records in the same order (tests/test_raw_batch.py pins the latter against the region query).
The device is tools/piece_standin.c, preloaded in front of libmdk_hip.so: device memory = host memory, k_inflate = zlib; everything
else -- device teams, the reorder buffer, device slabs taken member by member from their digests, reads carried across chunk boundaries
out of device members, chunks whose ranges point into device memory -- is the product's host code (csrc/host/mdk_io.c, mdk_pipeline.c)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

from conftest import synth

REPO = Path(__file__).resolve().parent.parent
STANDIN = REPO / "tools" / "_build" / "libmdk_piece_standin.so"
DRIVER = Path(__file__).resolve().parent / "hybrid_driver.py"


def chunks(args, attach, env=None):
    e = dict(os.environ)
    for k in ("MDK_DEVICE_INFLATE_ONLY", "MDK_GPU_PIECE_MB", "MDK_SLAB_CAP", "MDK_INFLATE_TEAMS", "MDK_HOST_INFLATE"):
        e.pop(k, None)
    e.update(env or {})
    if attach:
        e["LD_PRELOAD"] = str(STANDIN)
    r = subprocess.run([sys.executable, str(DRIVER), str(attach)] + [str(a) for a in args], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines()]
    return lines[:-1], lines[-1]["device_ranges"]


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    if not STANDIN.exists():
        subprocess.run(["make", "-C", str(REPO), "tools/_build/libmdk_piece_standin.so"], check=True, capture_output=True)
    d = tmp_path_factory.mktemp("hyb")
    synth(d / "s", "-L", "300000,80000", "-c", "25", "-s", "5", "--extras")          # two contigs, secondary/supplementary records sharing names
    synth(d / "x", "-L", "200000", "-c", "20", "-s", "6", "--split-records")
    return d


DEV = {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "0.25"}


@pytest.mark.parametrize("extra", [[], ["--chunkSize", "20000"], ["--chunkSize", "3333"], ["-r", "chrS1:50000-250000", "--chunkSize", "40000"]],
                         ids=["default", "chunk20k", "chunk3333", "region"])
@pytest.mark.parametrize("env", [DEV, dict(DEV, MDK_SLAB_CAP="2"), {"MDK_GPU_PIECE_MB": "0.25"}], ids=["device", "device-slabcap2", "mixed"])
def test_chunks_are_the_same_with_device_pieces(data, tmp_path, extra, env):
    base = [data / "s.fa", data / "s.bam"] + extra + ["-o", tmp_path / "o"]
    ref, _ = chunks(base, 0)
    got, n_dev = chunks(base, 1, env)
    assert got == ref and len(ref) >= 2 and sum(c.get("n", 0) for c in ref) > 10000
    if env.get("MDK_DEVICE_INFLATE_ONLY"):
        assert n_dev > 0


def test_split_records_with_device_pieces(data, tmp_path):
    base = [data / "x.fa", data / "x.bam", "--chunkSize", "15000", "-o", tmp_path / "o"]
    ref, _ = chunks(base, 0)
    got, _ = chunks(base, 1, DEV)
    assert got == ref and sum(c.get("n", 0) for c in ref) > 5000
