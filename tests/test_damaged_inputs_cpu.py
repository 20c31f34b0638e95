"""CPU: damaged BAM files through the command (on the device stand-in, tools/dev_standin.c: everything around the kernels is the product's
own code -- the BGZF framing under the file lock, the host inflate teams with their CRC32 check, the reader, the way an error travels to the
exit code) and through the oracle: the same exit code, no output where the oracle leaves none, identical output where it does, and never
a hang.  The reference reads through htslib, whose bgzf_read_block fails on a bad member, a CRC32 mismatch or a file that ends inside
a member, and `extract` leaves with the code of a file it cannot read (extract.c:1400-1404 for the open; a read error inside the iterator
ends the chunk's loop the same way)."""
import os
import subprocess

import pytest

import methyldackel_amd as mdk
from conftest import REPO, run_oracle, synth

STANDIN = REPO / "tools" / "_build" / "libmdk_dev_standin.so"


@pytest.fixture(scope="module")
def good(tmp_path_factory):
    if not STANDIN.exists():
        subprocess.run(["make", "-C", str(REPO), "tools/_build/libmdk_dev_standin.so"], check=True, capture_output=True)
    d = tmp_path_factory.mktemp("damaged")
    synth(d / "s", "-L", "150000,40000", "-c", "15", "-s", "97", "--extras")
    od = d / "oracle"; od.mkdir()
    r = run_oracle([str(d / "s.fa"), str(d / "s.bam"), "-o", "out"], cwd=od, dump=d / "dump.tsv")
    assert r.returncode == 0, r.stderr[-500:]
    return d


def damage(kind, data):
    n = len(data)
    if kind.startswith("flip"):
        b = bytearray(data); b[int(n * float(kind[4:]))] ^= 0x55; return bytes(b)
    if kind.startswith("cut"):
        v = kind[3:]; return data[:int(n * float(v)) if "." in v else (n + int(v) if v.startswith("-") else int(v))]
    return {"double_eof": data + data[-28:], "tail_garbage": data + b"xyz" * 10, "no_eof_block": data[:-28], "not_bgzf": b"this is not a BAM file\n" * 500}[kind]


@pytest.mark.parametrize("device_pieces", [False, True], ids=["host_inflate", "device_pieces"])
@pytest.mark.parametrize("kind", ["flip0.001", "flip0.3", "flip0.97", "cut0", "cut10", "cut28", "cut1000", "cut0.4", "cut-29", "cut-1", "double_eof", "tail_garbage", "no_eof_block", "not_bgzf"])
def test_damaged_bam_ends_as_in_the_oracle(good, tmp_path, kind, device_pieces):
    bam = tmp_path / "d.bam"
    bam.write_bytes(damage(kind, (good / "s.bam").read_bytes()))
    args = [str(good / "s.fa"), str(bam)]
    od = tmp_path / "oracle"; od.mkdir()
    o = run_oracle(args + ["-o", "out"], cwd=od)
    gd = tmp_path / "gpu"; gd.mkdir()
    env = {"LD_PRELOAD": str(STANDIN), "MDK_STANDIN_DUMP": str(good / "dump.tsv")}
    if device_pieces:       # test hooks (csrc/host/mdk_io.c): every piece after the header's goes through the device teams (md_piece_*: zlib in the stand-in), 256 KB each
        env.update({"MDK_DEVICE_INFLATE_ONLY": "1", "MDK_GPU_PIECE_MB": "0.25"})
    try:
        r = mdk.run_cli(args + ["-@", "4", "-o", "out"], cwd=gd, env=env, timeout=120)
    except subprocess.TimeoutExpired:
        pytest.fail(f"the command hangs on a BAM file damaged by {kind}")
    assert r.returncode == o.returncode, (kind, o.returncode, o.stderr[-300:], r.returncode, r.stderr[-600:])
    fo, fg = od / "out_CpG.bedGraph", gd / "out_CpG.bedGraph"
    if o.returncode == 0:
        assert fo.read_bytes() == fg.read_bytes()
    else:
        assert r.stderr.strip(), "an error exit says why"
        assert "d.bam" in r.stderr
