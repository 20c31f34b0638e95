"""CPU: the device inflate's decoder (csrc/mdk_inflate_core.h, compiled for the host) and the host re-statement of the kernel's
64-lane phases (tools/inflate_emu.cpp) against zlib -- deflate streams of every level and strategy (stored, fixed, dynamic blocks,
distance-1 runs, maximum-distance matches, empty input) and every BGZF member of the reference's fixture BAMs.  The kernel itself is
compared with zlib on the GPU (tests/test_gpu_inflate.py); this test keeps the shared decoder honest where there is no GPU."""
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
EMU = REPO / "tools" / "_build" / "inflate_emu"


@pytest.fixture(scope="module")
def emu():
    if not EMU.exists():
        subprocess.run(["make", "-C", str(REPO), "tools/_build/inflate_emu"], check=True, capture_output=True)
    return EMU


def test_damaged_streams_are_rejected_or_equal_zlib(emu):
    """bits flipped, bytes overwritten, streams cut short, wrong announced sizes: the decoder comes back, writes nothing beyond the announced
    size, and accepts only what zlib accepts, with the same bytes"""
    r = subprocess.run([str(emu), "--fuzz", "150000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 FAILURES" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_wavefront_crc32_equals_zlib(emu):
    """k_crc32's arithmetic (csrc/mdk_crc32_core.h: a register per lane over its column of the member, merged with GF(2) multiplications) on the
    host: every length up to 2100, random and maximal lengths, any alignment, all-zero / all-one / random bytes"""
    r = subprocess.run([str(emu), "--crc", "6000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '"mismatches": 0' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_selftest_streams(emu):
    r = subprocess.run([str(emu), "--selftest"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("bam", sorted(p.name for p in (REPO / "tests" / "golden").glob("*.bam")))
def test_fixture_members_equal_zlib(emu, bam):
    r = subprocess.run([str(emu), str(REPO / "tests" / "golden" / bam)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_synthetic_bam_members_equal_zlib(emu, small_synth):
    r = subprocess.run([str(emu), str(small_synth / "pe.bam")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
