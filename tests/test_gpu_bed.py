"""GPU parity for -l FILE / --keepStrand: the per-position region/strand restriction runs on the device
(k_mask_regions + the strand predicate in k_pileup); output files must be byte-identical to the oracle's."""
import filecmp
import os

import pytest

import methyldackel_amd as mdk
from bedgen import random_bed
from conftest import read_dump, run_oracle, synth
from test_gpu_parity import abi_sites, compare_cli

pytestmark = pytest.mark.gpu

PE = [("chrS1", 40000), ("chrS2", 20000)]
CLI = [
    (dict(n=40, seed=21), []),
    (dict(n=40, seed=21), ["--keepStrand"]),
    (dict(n=200, seed=22, crlf=True), ["--keepStrand", "--CHG", "--CHH"]),
    (dict(n=60, seed=23, gz=True), ["--keepStrand", "--chunkSize", "777", "--mergeContext", "--CHG"]),
    (dict(n=30, seed=24, dense=True), ["--keepStrand", "--cytosine_report", "--CHH", "--chunkSize", "5000"]),
    (dict(n=10, seed=25), ["--keepStrand", "-r", "chrS1:5000-30000", "--methylKit"]),
    (dict(n=80, seed=26), ["--keepStrand", "--minOppositeDepth", "2", "--maxVariantFrac", "0.3", "--mergeContext"]),
    (dict(n=80, seed=27), ["--keepStrand", "--OT", "5,90,5,90", "--nOB", "3,3,3,3", "-p", "15", "--fraction"]),
    (dict(n=50, seed=28), ["--keepStrand", "-@", "4", "--chunkSize", "2500", "--counts", "--CHG"]),
]


@pytest.mark.parametrize("bk,extra", CLI, ids=[f"n{b['n']}s{b['seed']}:{' '.join(e)}" for b, e in CLI])
@pytest.mark.parametrize("env", [None, {"MDK_TILE": "512"}], ids=["tile-default", "tile-512"])
def test_cli_bed_byte_exact(tmp_path, small_synth, bk, extra, env):
    bed = random_bed(tmp_path / ("r.bed.gz" if bk.get("gz") else "r.bed"), PE, **bk)
    compare_cli(tmp_path, [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-l", str(bed)] + extra, env=env)


def test_cli_bed_without_index_byte_exact(tmp_path, small_synth):
    bed = random_bed(tmp_path / "r.bed", PE, n=12, seed=29)
    compare_cli(tmp_path, [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-l", str(bed), "--keepStrand", "--chunkSize", "1000"], env={"MDK_NO_INDEX": "1"})


def test_abi_sites_with_regions_equal_oracle_counters(tmp_path, small_synth):
    bed = random_bed(tmp_path / "r.bed", PE, n=120, seed=30)
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-l", str(bed), "--keepStrand", "--CHG", "--CHH", "--minOppositeDepth", "1", "--chunkSize", "9000"]
    dump = tmp_path / "d.tsv"
    assert run_oracle(args + ["-o", tmp_path / "o"], cwd=tmp_path, dump=dump).returncode == 0
    assert abi_sites(args + ["-o", tmp_path / "g"]) == read_dump(dump)


def test_set_regions_rejects_bad_runs(tmp_path, small_synth):
    plan = mdk.Plan([str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-o", str(tmp_path / "x")])
    dev = mdk.Device(plan.dev_cfg())
    with pytest.raises(mdk.MdkError):
        dev.set_regions(0, [(0, 10, 0)])                      # reference not uploaded yet
    plan.ensure_reference(dev, 0)
    for bad in ([(10, 5, 0)], [(0, 10, 0), (5, 20, 0)], [(0, 10, 3)], [(-1, 10, 0)]):
        with pytest.raises(mdk.MdkError):
            dev.set_regions(0, bad)
    dev.set_regions(0, [(0, 10, 0), (10, 20, 2)])
    dev.set_regions(0, [])
    dev.close(); plan.close()


def test_sharded_two_ranks_with_bed_byte_exact(tmp_path, small_synth):
    """-l with the command as two processes: chunks no region touches are passed over by every rank"""
    bed = random_bed(tmp_path / "r.bed", PE, n=25, seed=31)
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-l", str(bed), "--keepStrand", "--CHG", "--chunkSize", "3000"]
    compare_cli(tmp_path, args, ranks=2, env={"MDK_DEVICE": "0"})


def test_bed_1mb_panel_byte_exact(tmp_path):
    """a capture-panel-like BED (300 short targets) over the 1 Mb / 30x sample"""
    synth(tmp_path / "S", "-L", "1000000", "-c", "30", "-s", "77")
    bed = random_bed(tmp_path / "panel.bed", [("chrS1", 1000000)], n=300, seed=32, max_len=600)
    compare_cli(tmp_path, [str(tmp_path / "S.fa"), str(tmp_path / "S.bam"), "-l", str(bed), "--keepStrand", "-@", "8"])
