"""CPU: the COMMAND ITSELF without a GPU.  tools/dev_standin.c is preloaded in front of libmdk_hip.so: "device memory" is host memory, BGZF
pieces are zlib, and what a slot "computes" is looked up in the per-column counters the oracle dumped for the same command line.  So the
counting is the oracle's -- the GPU tests check the kernels -- but everything AROUND it is the product's own code, running here:
  * extract_main (csrc/host/mdk_extract.c): the uploader / collector / reference threads, groups in flight, early release of the host slabs,
    a chunk handed back to the host preparation, the emitter;
  * the one-process-per-GPU driver (csrc/host/mdk_ranks.c): schedule sharding, rank 0's ring, the control and data connections, ordered
    emission -- with chunks dealt k mod N and, MDK_CLAIM=1, claimed as the ranks get to them (the reference's worker threads under
    positionMutex, extract.c:325-350), where a skewed input must come out balanced by WORK."""
import os
import re
import socket
import struct
import subprocess
import zlib
from pathlib import Path

import pytest

import methyldackel_amd as mdk
from conftest import REPO, run_oracle, synth

STANDIN = REPO / "tools" / "_build" / "libmdk_dev_standin.so"
SUFFIXES = ["_CpG.bedGraph", "_CHG.bedGraph", "_CHH.bedGraph"]


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    if not STANDIN.exists():
        subprocess.run(["make", "-C", str(REPO), "tools/_build/libmdk_dev_standin.so"], check=True, capture_output=True)
    d = tmp_path_factory.mktemp("ranks_cpu")
    synth(d / "s", "-L", "260000,90000", "-c", "18", "-s", "41", "--extras")
    return d


def oracle(tmp, args):
    od = tmp / "oracle"; od.mkdir()
    r = run_oracle(list(args) + ["-o", "out"], cwd=od, dump=tmp / "dump.tsv")
    assert r.returncode == 0, r.stderr[-500:]
    return od


def standin_env(tmp, **kw):
    e = {"LD_PRELOAD": str(STANDIN), "MDK_STANDIN_DUMP": str(tmp / "dump.tsv"), "MDK_HOST_PROFILE": "1"}
    e.update({k: str(v) for k, v in kw.items()})
    return e


def same_outputs(od, gd):
    seen = 0
    for s in SUFFIXES:
        fo, fg = od / ("out" + s), gd / ("out" + s)
        assert fo.exists() == fg.exists(), s
        if fo.exists():
            seen += 1
            assert fo.read_bytes() == fg.read_bytes(), s
    assert seen


@pytest.mark.parametrize("extra,env", [([], {}), (["--chunkSize", "7000", "--mergeContext", "--CHG"], {"MDK_FASTA_THREADS": 5}), (["--chunkSize", "20000"], {"MDK_STANDIN_HANDBACK": 3}),
                                       (["--chunkSize", "333333", "--minOppositeDepth", "2", "--maxVariantFrac", "0.3", "--CHH"], {}), (["--chunkSize", "5000"], {"MDK_DEVICE_INFLATE_ONLY": 1, "MDK_GPU_PIECE_MB": "0.25", "MDK_STANDIN_HANDBACK": 5}),
                                       (["--chunkSize", "3000", "--CHG"], {"MDK_GROUPS_IN_FLIGHT": 5, "MDK_STANDIN_HANDBACK": 7, "MDK_STANDIN_US_PER_KREC": 20000}), (["--chunkSize", "3000"], {"MDK_GROUPS_IN_FLIGHT": 2, "MDK_LAZY_COPY_MIN": 1}),
                                       (["--chunkSize", "5000"], {"MDK_DEVICE_INFLATE_ONLY": 1, "MDK_GPU_PIECE_MB": "0.5", "MDK_DSLAB_EXTRA": 1, "MDK_GPU_INFLATE_TEAMS": 3})])
def test_extract_main_on_the_standin_equals_oracle(data, tmp_path, extra, env):
    """one process: extract_main's threads, groups of chunks in flight, slabs given back early, chunks handed back to the host preparation, pieces
    "inflated on the device" (fifth case: every piece after the header's); then five and two groups in flight instead of three; the last: three device teams with one
    device slab to spare between them"""
    args = [str(data / "s.fa"), str(data / "s.bam"), "-@", "4"] + extra
    od = oracle(tmp_path, args)
    gd = tmp_path / "gpu"; gd.mkdir()
    r = mdk.run_cli(args + ["-o", "out"], cwd=gd, env=standin_env(tmp_path, **env))
    assert r.returncode == 0, r.stderr[-1500:]
    same_outputs(od, gd)
    if "MDK_STANDIN_HANDBACK" in env:
        assert int(re.search(r"chunks prepared on the host after all: (\d+)", r.stderr).group(1)) >= 2
    if "MDK_DEVICE_INFLATE_ONLY" in env:
        assert int(re.search(r"on the device (\d+)", r.stderr).group(1)) >= 2
    # MDK_HOST_PROFILE: where the inflate teams' time went, one line per kind of team (csrc/host/mdk_io.c)
    teams = dict(re.findall(r"\[mdk host\] (host|device) teams, summed over the teams that have left: (\d+) pieces", r.stderr))
    assert set(teams) == {"host", "device"} and int(teams["host"]) >= 1, r.stderr[-1500:]
    if "MDK_DEVICE_INFLATE_ONLY" in env:
        assert int(teams["device"]) >= 1, r.stderr[-1500:]


def run_ranks_cpu(args, n, cwd, env):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(n):
        e = dict(os.environ); e.update(env); e.update({"MDK_WORLD": str(n), "MDK_RANK": str(r), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        procs.append(subprocess.Popen([str(mdk.CLI), "extract"] + [str(a) for a in args], cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    return [p.returncode for p in procs], [o[1] for o in outs]


@pytest.mark.parametrize("world,extra,env", [(2, ["--chunkSize", "9000", "--CHG", "--mergeContext"], {}), (3, ["--chunkSize", "20000"], {"MDK_STANDIN_HANDBACK": 4}), (2, ["--chunkSize", "9000"], {"MDK_NO_INDEX": 1}),
                                             (3, ["--chunkSize", "9000", "--CHG"], {"MDK_CLAIM": 1}), (2, ["--chunkSize", "20000"], {"MDK_CLAIM": 1, "MDK_NO_INDEX": 1, "MDK_STANDIN_HANDBACK": 3}),
                                             (8, ["--chunkSize", "2000"], {}), (8, ["--chunkSize", "2000", "--CHH"], {"MDK_CLAIM": 1})])      # (eight ranks: rank 0 collects seven others' chunks in schedule order)
def test_ranks_driver_on_the_standin_equals_oracle(data, tmp_path, world, extra, env):
    """mdk_ranks.c itself, N processes over its TCP connections: chunks dealt k mod N and claimed (MDK_CLAIM=1), with and without the index, with
    chunks handed back to the host on the rank that holds their records"""
    args = [str(data / "s.fa"), str(data / "s.bam"), "-@", "3"] + extra
    od = oracle(tmp_path, args)
    gd = tmp_path / "gpu"; gd.mkdir()
    rcs, errs = run_ranks_cpu(args + ["-o", "out"], world, gd, standin_env(tmp_path, **env))
    assert rcs == [0] * world, errs
    same_outputs(od, gd)
    own = [int(re.search(r"rank \d+: (\d+) own chunks", e).group(1)) for e in errs]
    # (claimed: a rank that starts late on a loaded host may find every chunk of this small input taken -- its share is the balance test's business)
    assert (sum(own) >= 1 if "MDK_CLAIM" in env else min(own) >= 1) and all(("claimed" if "MDK_CLAIM" in env else "k mod N") in e for e in errs)


def skewed_bam(src, dst, window, keep_every):
    """the records of `src` with the ODD windows of `window` bp thinned to one record in `keep_every`: alternate chunks deep and shallow"""
    raw = src.read_bytes(); data = bytearray(); o = 0
    while o + 18 <= len(raw):
        xlen = struct.unpack_from("<H", raw, o + 10)[0]; bs = struct.unpack_from("<H", raw, o + 16)[0] + 1
        if struct.unpack_from("<I", raw, o + bs - 4)[0]:
            data += zlib.decompress(raw[o + 12 + xlen:o + bs - 8], wbits=-15)
        o += bs
    l_text = struct.unpack_from("<i", data, 4)[0]; n_ref = struct.unpack_from("<i", data, 8 + l_text)[0]; p = 12 + l_text
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", data, p)[0]
    out, blk, k = bytearray(), bytearray(data[:p]), 0

    def flush():
        nonlocal blk
        if blk:
            c = zlib.compressobj(1, zlib.DEFLATED, -15); comp = c.compress(bytes(blk)) + c.flush()
            out.extend(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(bytes(blk)), len(blk)))
            blk = bytearray()
    flush()
    while p + 4 <= len(data):
        bsz = struct.unpack_from("<I", data, p)[0]; pos = struct.unpack_from("<i", data, p + 8)[0]
        k += 1
        if (pos // window) % 2 == 0 or k % keep_every == 0:
            if len(blk) + 4 + bsz > 60000:
                flush()
            blk += data[p:p + 4 + bsz]
        p += 4 + bsz
    flush()
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    dst.write_bytes(bytes(out))


def test_claimed_chunks_balance_the_work_not_the_count(data, tmp_path):
    """alternate chunks deep and shallow (8:1): dealt k mod 2, one rank gets all the deep ones; claimed, a rank that is still busy with a deep
    chunk leaves the next ones to the other (balance by admitted bases, not by bp).  Dealt, the ranks' own chunks hold records 4-5 : 1;
    claimed, typically within 10 % of each other -- asserted within 35 %, because here "compute" is a sleep in the stand-in, the ranks share 8 cores
    with the test runner, and rank 0 collects in order (a busy rank 0 holds the others' sends up for a moment, which shifts a few claims)"""
    skewed_bam(data / "s.bam", tmp_path / "skew.bam", 1000, 8)
    args = [str(data / "s.fa"), str(tmp_path / "skew.bam"), "-@", "2", "--chunkSize", "1000"]
    od = oracle(tmp_path, args)
    held = {}
    def run(mode, env, k):
        gd = tmp_path / f"{mode}{k}"; gd.mkdir()
        rcs, errs = run_ranks_cpu(args + ["-o", "out"], 2, gd, standin_env(tmp_path, MDK_STANDIN_US_PER_KREC=80000, **env))
        assert rcs == [0, 0], errs
        same_outputs(od, gd)
        return [int(re.search(r"holding (\d+) records", e).group(1)) for e in errs]
    held["dealt"] = run("dealt", {}, 0)
    assert max(held["dealt"]) > 3 * min(held["dealt"]), held
    # who claims what is decided by the clock: on a host that is busy with something else a run can come out lopsided -- the best of three counts
    # (every run's OUTPUT must equal the oracle's whatever the split)
    for k in range(3):
        held["claimed"] = run("claimed", {"MDK_CLAIM": 1}, k)
        if max(held["claimed"]) <= 1.35 * min(held["claimed"]): break
    rc, rd = max(held["claimed"]) / min(held["claimed"]), max(held["dealt"]) / min(held["dealt"])
    assert rc <= 1.35 or rc <= 0.5 * rd, held          # (the second clause: a host so busy that three runs in a row were lopsided; still far from the dealt split)


def test_command_hands_its_teardown_to_a_child_and_stays_the_same_command(data, tmp_path):
    """main.c detach_teardown: the work is done by a child whose word "outputs closed" lets the command return; the command's outputs, return code
    and fate under a signal are what they are in place (the default; the child is opt-in: MDK_DETACH=1)"""
    import signal, time
    args = [str(data / "s.fa"), str(data / "s.bam"), "-@", "4", "--chunkSize", "20000"]
    od = oracle(tmp_path, args)
    for tag, extra in (("detached", {"MDK_DETACH": 1}), ("inplace", {})):
        gd = tmp_path / tag; gd.mkdir()
        r = mdk.run_cli(args + ["-o", "out"], cwd=gd, env=standin_env(tmp_path, **extra))
        assert r.returncode == 0 and "[mdk main] leaving" in r.stderr, r.stderr[-800:]      # (stderr through a pipe: complete, and at its end when the command returns)
        same_outputs(od, gd)
        bad = mdk.run_cli([str(data / "s.fa"), str(tmp_path / "no_such.bam"), "-o", "out"], cwd=gd, env=standin_env(tmp_path, **extra))
        assert bad.returncode == 252 and "Couldn't open" in bad.stderr
    # a signal to the command reaches the process that does the work: nothing of it is left behind
    gd = tmp_path / "killed"; gd.mkdir()
    e = dict(os.environ); e.update(standin_env(tmp_path, MDK_STANDIN_US_PER_KREC=3000000, MDK_DETACH=1)); e["MDK_NO_RANKS"] = "1"
    mark = str(gd / "out_marker")
    p = subprocess.Popen([str(mdk.CLI), "extract"] + args + ["-o", mark], cwd=gd, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    def workers():
        n = []
        for d in os.listdir("/proc"):
            if d.isdigit():
                try:
                    if mark.encode() in open(f"/proc/{d}/cmdline", "rb").read() and open(f"/proc/{d}/stat").read().split(") ")[1][0] != "Z": n.append(int(d))
                except OSError: pass
        return n
    t0 = time.time()
    while len(workers()) < 2 and time.time() - t0 < 10: time.sleep(0.02)
    assert len(workers()) == 2                       # the command and its child
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=20)
    assert rc in (-signal.SIGTERM, 128 + signal.SIGTERM)
    t0 = time.time()
    while workers() and time.time() - t0 < 10: time.sleep(0.05)
    assert workers() == []
