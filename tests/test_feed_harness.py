"""CPU: the BAM feed of the host pipeline -- host inflate teams, "device" inflate teams, the reorder buffer, slab accounting
(csrc/host/mdk_io.c) -- without a GPU.  tools/feed_harness.c compiles the reader in as it is and replaces what it calls in the device
library by stand-ins (device memory = host memory, k_inflate = zlib, k_walk = the host's record walk), reads every record in stream order
and prints an order-sensitive digest.  Every configuration must give the digest of the host-only pass, and must END: the last cases hold
more slabs than the inflaters may allocate ahead -- with the piece counter shared by both kinds of team that once stopped the 128 Mb run
(a team holding the piece the scanner needed next waited for a slab while out-of-order deliveries made the scanner look busy)."""
import json
import subprocess
from pathlib import Path

import pytest

from conftest import synth

REPO = Path(__file__).resolve().parent.parent
HARNESS = REPO / "tools" / "_build" / "feed_harness"


def run(bam, mode, hold, threads=8, env=None, timeout=120):
    import os
    e = dict(os.environ)
    for k in ("MDK_DEVICE_INFLATE_ONLY", "MDK_GPU_PIECE_MB", "MDK_SLAB_CAP", "MDK_INFLATE_TEAMS"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([str(HARNESS), str(bam), str(mode), str(hold), str(threads)], capture_output=True, text=True, env=e, timeout=timeout)
    assert r.returncode == 0, r.stderr[-1000:]
    return json.loads(r.stdout)


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    if not HARNESS.exists():
        subprocess.run(["make", "-C", str(REPO), "tools/_build/feed_harness"], check=True, capture_output=True)
    d = tmp_path_factory.mktemp("feed")
    synth(d / "s", "-L", "1500000", "-c", "30", "-s", "5")                 # ~25 MB of BAM: three host pieces, or a hundred small device pieces
    synth(d / "x", "-L", "500000", "-c", "20", "-s", "6", "--split-records")   # records that straddle BGZF members: device slabs are read back
    return d


CONFIGS = [
    (1, 0, 8, {}),                                                                                          # hybrid, defaults
    (1, 6, 32, {}),                                                                                         # four host teams + three device teams
    (1, 0, 8, {"MDK_GPU_PIECE_MB": "0.25"}),                                                               # many small device pieces among host pieces
    (1, 12, 8, {"MDK_GPU_PIECE_MB": "0.25", "MDK_SLAB_CAP": "2"}),                                         # the consumer holds more slabs than the cap
    (1, 12, 8, {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "0.25", "MDK_SLAB_CAP": "2"}),        # ... every piece after the header on the "device"
    (1, 40, 8, {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "0.25"}),
    (0, 12, 8, {"MDK_SLAB_CAP": "2", "MDK_INFLATE_TEAMS": "4"}),                                            # host teams only, delivering out of order
    (1, 3, 1, {"MDK_GPU_PIECE_MB": "0.5"}),                                                                # -@ 1: one host thread next to the device teams
    (1, 20, 2, {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "1", "MDK_SLAB_CAP": "2"}),
    (0, 0, 1, {}),
]


@pytest.mark.parametrize("mode,hold,threads,env", CONFIGS, ids=[f"mode{m}-hold{h}-t{t}-" + ",".join(f"{k[4:]}={v}" for k, v in e.items()) for m, h, t, e in CONFIGS])
def test_every_configuration_reads_the_same_stream(sample, mode, hold, threads, env):
    ref = run(sample / "s.bam", 0, 0)
    got = run(sample / "s.bam", mode, hold, threads, env)
    assert (got["records"], got["bytes"], got["digest"]) == (ref["records"], ref["bytes"], ref["digest"])
    assert ref["records"] > 100_000
    if env.get("MDK_DEVICE_INFLATE_ONLY"):
        assert got["device_pieces"] > 20 and got["host_pieces"] <= 8 and got["read_back"] == 0


def test_split_records_are_read_back_and_still_agree(sample):
    ref = run(sample / "x.bam", 0, 0)
    got = run(sample / "x.bam", 1, 3, 8, {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "0.25"})
    assert (got["records"], got["digest"]) == (ref["records"], ref["digest"]) and ref["records"] > 1000
    assert got["device_pieces"] > 0 and got["read_back"] > 0          # a slab whose members do not start on record boundaries comes back to the host


def test_fixture_bams(sample):
    for bam in sorted((REPO / "tests" / "golden").glob("*.bam")):
        ref = run(bam, 0, 0)
        got = run(bam, 1, 2, 8, {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "0.25"})
        assert (got["records"], got["digest"]) == (ref["records"], ref["digest"]), bam.name


# ---- pieces cut without the file's lock (mdk_io.c claim_range / frame_range; MDK_SPEC_FRAMING=1 -- measured slower end to end, not the default --: from the moment the device is attached; with MDK_SPEC_AT_ONCE=1 from the file's first byte) ----
SPEC = [
    (1, 6, 32, {}),
    (1, 12, 8, {"MDK_GPU_PIECE_MB": "0.25", "MDK_SLAB_CAP": "2"}),
    (0, 12, 8, {"MDK_SLAB_CAP": "2", "MDK_INFLATE_TEAMS": "4"}),
    (1, 40, 8, {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "0.25"}),
]


@pytest.mark.parametrize("mode,hold,threads,env", SPEC, ids=[f"spec{i}" for i in range(len(SPEC))])
def test_pieces_cut_without_the_lock_read_the_same_stream_as_the_serial_walk(sample, mode, hold, threads, env):
    for bam in ("s.bam", "x.bam"):
        ref = run(sample / bam, mode, hold, threads, dict(env, MDK_SERIAL_FRAMING="1"))
        got = run(sample / bam, mode, hold, threads, dict(env, MDK_SPEC_FRAMING="1", MDK_SPEC_AT_ONCE="1"))
        mapped = run(sample / bam, mode, hold, threads, dict(env, MDK_SPEC_FRAMING="1", MDK_SPEC_AT_ONCE="1", MDK_SPEC_PREAD="1"))
        switch = run(sample / bam, mode, hold, threads, dict(env, MDK_SPEC_FRAMING="1"))          # (cut by the walk until the device is attached, without the lock from then on)
        assert (switch["records"], switch["bytes"], switch["digest"]) == (ref["records"], ref["bytes"], ref["digest"]) and switch["spec_redo"] == 0
        assert (got["records"], got["bytes"], got["digest"]) == (ref["records"], ref["bytes"], ref["digest"]) == (mapped["records"], mapped["bytes"], mapped["digest"])
        assert got.get("spec_redo", 0) == 0 and ref.get("spec_redo", 0) == 0


@pytest.mark.parametrize("fault", [0, 1, 2, 5, 17, 60])
def test_a_piece_that_does_not_fit_sends_the_rest_of_the_file_down_the_serial_path(sample, fault):
    """MDK_SPEC_FAULT=n: piece n begins one member late -- the scanner must notice (it does not begin where piece n-1 ended), throw away
    what the teams hold and have the rest framed under the lock; the stream read is the same, and the run ends."""
    ref = run(sample / "s.bam", 0, 0, 8, {"MDK_SERIAL_FRAMING": "1"})
    for mode, hold, threads, env in ((1, 6, 8, {"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "0.25"}), (0, 3, 8, {"MDK_INFLATE_TEAMS": "4"})):
        got = run(sample / "s.bam", mode, hold, threads, dict(env, MDK_SPEC_FRAMING="1", MDK_SPEC_AT_ONCE="1", MDK_SPEC_FAULT=str(fault)))
        assert (got["records"], got["bytes"], got["digest"]) == (ref["records"], ref["bytes"], ref["digest"])
        if mode == 1 or fault <= 2:          # (the 25 MB file is three or four host pieces, or a hundred small device pieces: a piece number it does not reach meets no fault)
            assert got["spec_redo"] == 1
