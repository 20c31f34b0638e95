"""GPU parity for `mbias`: the histogram kernel (k_mbias) behind `MethylDackel mbias` against the oracle's restatement
of MBias.c/svg.c -- table, suggestion line and SVG files byte-identical; and at the C-ABI (md_dev_mbias_submit/read)."""
import filecmp

import numpy as np
import pytest

import methyldackel_amd as mdk
from bedgen import random_bed
from conftest import GOLDEN, synth
from test_mbias import FIX, SYN, oracle_mbias, parse_txt

pytestmark = pytest.mark.gpu


def compare_mbias(tmp_path, args, env=None, svg=True):
    od, gd = tmp_path / "oracle", tmp_path / "gpu"
    od.mkdir(exist_ok=True), gd.mkdir(exist_ok=True)
    tail = (["out"] if svg else ["--noSVG"]) + ["--txt"]
    ro = oracle_mbias(list(args) + tail, cwd=od)
    rg = mdk.run_cli(list(args) + tail, cwd=gd, env=env, command="mbias")
    assert rg.returncode == ro.returncode, (rg.returncode, ro.returncode, rg.stderr[-2000:])
    assert rg.stdout == ro.stdout
    sug = lambda s: [l for l in s.splitlines() if l.startswith("Suggested")]
    assert sug(rg.stderr) == sug(ro.stderr)
    so, sg = sorted(f.name for f in od.iterdir() if f.suffix == ".svg"), sorted(f.name for f in gd.iterdir() if f.suffix == ".svg")
    assert so == sg and (bool(so) == svg or not parse_txt(ro.stdout))
    for f in so:
        assert filecmp.cmp(od / f, gd / f, shallow=False), f
    return ro


@pytest.mark.parametrize("args", FIX, ids=[" ".join(a[1:]).replace(str(GOLDEN) + "/", "") for a in FIX])
def test_cli_fixtures(tmp_path, args):
    compare_mbias(tmp_path, args)


@pytest.mark.parametrize("which,extra", SYN, ids=[f"{w}:{' '.join(e)}" for w, e in SYN])
@pytest.mark.parametrize("env", [None, {"MDK_TILE": "512"}], ids=["tile-default", "tile-512"])
def test_cli_synthetic(tmp_path, small_synth, which, extra, env):
    compare_mbias(tmp_path, [str(small_synth / f"{which}.fa"), str(small_synth / f"{which}.bam")] + extra, env=env)


def test_cli_bed_and_threads(tmp_path, small_synth):
    bed = random_bed(tmp_path / "r.bed", [("chrS1", 40000), ("chrS2", 20000)], n=60, seed=61)
    compare_mbias(tmp_path, [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-l", str(bed), "--keepStrand", "--CHG", "--chunkSize", "2500", "-@", "4"], svg=False)


def test_abi_histogram_equals_oracle_table(tmp_path, small_synth):
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--CHG", "--CHH", "--chunkSize", "4000", "--nOT", "2,3,4,5", "--noSVG"]
    o = oracle_mbias(args, cwd=tmp_path)
    assert o.returncode == 0
    want = parse_txt(o.stdout)
    plan = mdk.Plan(args, command="mbias")
    dev = mdk.Device(plan.dev_cfg())
    k, keep = 0, {}
    while True:
        dev.slot_sync(k & 1)
        c = plan.next_chunk()
        if c is None:
            break
        if c.skipped:
            continue
        plan.ensure_reference(dev, c.tid)
        keep[k & 1] = c
        dev.mbias_submit(k & 1, c.batch)
        k += 1
    h = dev.mbias_read()
    got = {(s + 1, r + 1, q): [int(h[q, s, r, 0]), int(h[q, s, r, 1])] for q in range(h.shape[0]) for s in range(4) for r in range(2) if h[q, s, r].any()}
    assert got == want
    dev.mbias_reset()
    assert dev.mbias_read().shape[0] == 0
    dev.close(); plan.close()


def test_long_reads_spill_past_the_lds_rows(tmp_path):
    """reads longer than the 512 histogram rows a workgroup keeps in LDS: the tail goes to the global histogram directly"""
    synth(tmp_path / "L", "-L", "60000", "-c", "12", "-l", "700", "-s", "91", "--single")
    compare_mbias(tmp_path, [str(tmp_path / "L.fa"), str(tmp_path / "L.bam"), "--CHG", "--CHH"])


def test_s1_mbias(tmp_path):
    synth(tmp_path / "S1", "-L", "1000000", "-c", "30", "-s", "0x5EED0001")
    compare_mbias(tmp_path, [str(tmp_path / "S1.fa"), str(tmp_path / "S1.bam"), "-@", "8"])
    compare_mbias(tmp_path, [str(tmp_path / "S1.fa"), str(tmp_path / "S1.bam"), "-@", "8", "--CHG", "--CHH", "--nOT", "6,6,6,6", "--nOB", "6,6,6,6"])


@pytest.mark.parametrize("extra", [["--CHG", "--CHH", "--chunkSize", "4000", "--nOT", "2,3,4,5"], ["-q", "0", "-F", "0", "--keepDupes", "--keepSingleton", "--keepDiscordant", "--chunkSize", "9000"],
                                   ["--noCpG", "--CHH", "-p", "20", "--nOB", "1,2,3,4"], ["--CHH", "--minConversionEfficiency", "0.6", "--chunkSize", "5000"]], ids=["allctx", "everything_admitted", "chh_trim", "conversion_efficiency"])
def test_abi_histogram_device_prep_equals_host_prep(tmp_path, small_synth, extra):
    """md_dev_mbias_submit_raw (records prepared on the device, no pairing) accumulates the same histogram as md_dev_mbias_submit
    of the host-built batches; the command (which uses the raw path) is compared with the oracle by the tests above, and with
    MDK_HOST_PREP=1 below"""
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam")] + extra + ["--noSVG"]
    hists = []
    for mode in (0, 1):
        plan = mdk.Plan(args, command="mbias")
        plan.set_prep(mode)
        dev = mdk.Device(plan.dev_cfg())
        if mode:
            dev.set_prep(plan.prep_cfg())
        k = 0
        while True:
            dev.slot_sync(k & 1)
            c = plan.next_chunk()
            if c is None:
                break
            if c.skipped:
                continue
            plan.ensure_reference(dev, c.tid)
            if mode:
                assert c.prep == 1
                dev.mbias_submit_raw(k & 1, c.raw)
            else:
                dev.mbias_submit(k & 1, c.batch)
            k += 1
        hists.append(dev.mbias_read())
        dev.close(); plan.close()
    assert hists[0].shape == hists[1].shape and (hists[0] == hists[1]).all() and hists[0].sum() > 1000


def test_cli_host_prep_mode(tmp_path, small_synth):
    compare_mbias(tmp_path, [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--CHG", "--chunkSize", "7000"], env={"MDK_HOST_PREP": "1"}, svg=False)
