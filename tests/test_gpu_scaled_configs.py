"""BASELINE.json configs[3] and configs[4] at a stated scale factor ("run at a stated scale factor ... report
extrapolation separately"), byte-exact against the oracle:

  configs[3]  hg38-scale 3 Gb, 30x, CpG extract sharded by contig/interval over the GPUs
              -> 24 contigs whose lengths are hg38's chr1..22,X,Y divided by 62 (49.4 Mb in total, 0.7-4.0 Mb each), 30x,
                 default 1 Mb chunking, one GPU and a 2-rank interval-sharded run.  Scale factor 1/62.
  configs[4]  3 Gb, 100x, --mergeContext with the -M bigWig mappability filter
              -> one 8 Mb contig, 100x, `--mergeContext -M map.bw` (the oracle reads the same track as BBM).  Scale 1/375.
"""
import filecmp
import os
import re
import socket
import subprocess

import pytest

import methyldackel_amd as mdk
from conftest import ORACLE, synth

pytestmark = pytest.mark.gpu

# hg38 primary assembly lengths (Mb, rounded) / 62
HG38_MB = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]
LADDER = [int(x * 1e6 / 62) for x in HG38_MB]


def oracle_run(args, cwd, threads):
    cwd.mkdir(exist_ok=True)
    r = subprocess.run([str(ORACLE), "extract"] + [str(a) for a in args] + ["-@", str(threads), "-o", "out"], cwd=cwd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


def same_outputs(od, gd):
    names = sorted(f for f in os.listdir(od) if f.startswith("out"))
    assert names
    for f in names:
        assert filecmp.cmp(od / f, gd / f, shallow=False), f
    return names


@pytest.fixture(scope="module")
def ladder(tmp_path_factory):
    d = tmp_path_factory.mktemp("ladder")
    synth(d / "g", "-L", ",".join(str(x) for x in LADDER), "-c", "30", "-s", "303", "-z", "1")
    oracle_run([d / "g.fa", d / "g.bam"], d / "oracle", os.cpu_count() or 8)
    return d


def test_config3_contig_ladder_one_gpu(ladder, tmp_path):
    """24 contigs, 49.4 Mb, 30x, default options: `MethylDackel extract` == oracle, byte for byte"""
    gd = tmp_path / "gpu"; gd.mkdir()
    r = mdk.run_cli([str(ladder / "g.fa"), str(ladder / "g.bam"), "-@", "32", "-o", "out"], cwd=gd)
    assert r.returncode == 0, r.stderr[-2000:]
    same_outputs(ladder / "oracle", gd)
    n = sum(1 for _ in open(gd / "out_CpG.bedGraph"))
    assert n > 900_000        # ~1 CpG per 50 bp and strand, almost all covered at 30x


@pytest.mark.parametrize("n", [2, 4])
def test_config3_contig_ladder_ranks(ladder, tmp_path, n):
    """the same run as N processes (all on this box's GPU): chunk k belongs to rank k % N, rank 0 collects and writes; 24
    contigs exercise the contig hand-over of the schedule on every rank"""
    gd = tmp_path / "gpu"; gd.mkdir()
    r = mdk.run_cli([str(ladder / "g.fa"), str(ladder / "g.bam"), "-@", "16", "-o", "out"], cwd=gd, ranks=n, env={"MDK_DEVICE": "0", "MDK_HOST_PROFILE": "1"})
    assert r.rank_returncodes == [0] * n, r.rank_stderr
    own = [int(re.search(r"rank %d: (\d+) own chunks" % k, r.rank_stderr[k]).group(1)) for k in range(n)]
    assert min(own) > 10 and max(own) - min(own) <= 1
    same_outputs(ladder / "oracle", gd)


def test_config4_100x_merge_bigwig(tmp_path):
    """8 Mb at 100x with --mergeContext and the bigWig mappability filter (-M, own bbi reader); the oracle has no bigWig
    reader and is given the same track as BBM (-B), which the product also accepts and must agree with"""
    synth(tmp_path / "d", "-L", "8000000", "-c", "100", "-s", "404", "-z", "1", "--bbm", "--bw")
    fa, bam = tmp_path / "d.fa", tmp_path / "d.bam"
    oracle_run([fa, bam, "--mergeContext", "-B", tmp_path / "d.bbm"], tmp_path / "oracle", os.cpu_count() or 8)
    for flag, track in (("-M", "d.bw"), ("-B", "d.bbm")):
        gd = tmp_path / ("gpu" + flag); gd.mkdir()
        r = mdk.run_cli([str(fa), str(bam), "--mergeContext", flag, str(tmp_path / track), "-@", "32", "-o", "out"], cwd=gd)
        assert r.returncode == 0, r.stderr[-2000:]
        same_outputs(tmp_path / "oracle", gd)
    assert sum(1 for _ in open(tmp_path / "oracle" / "out_CpG.bedGraph")) > 50_000
