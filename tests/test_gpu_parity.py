"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C-ABI of the product:
  * CLI level: `MethylDackel extract` (extract_main -> libmdk_hip.so kernels) must produce byte-identical output files
    and stdout to the CPU oracle for the same command line;
  * C-ABI level: md_dev_submit/md_dev_download sites == the oracle's per-column counters.
Bar: bit-exact (integer counters, text formatted by the same libc)."""
import ctypes as C
import filecmp
import os
import subprocess

import pytest

import methyldackel_amd as mdk
from conftest import GOLDEN, read_dump, run_oracle, synth

pytestmark = pytest.mark.gpu

SUFFIXES = ["_CpG.bedGraph", "_CHG.bedGraph", "_CHH.bedGraph", "_CpG.meth.bedGraph", "_CHG.meth.bedGraph", "_CHH.meth.bedGraph",
            "_CpG.counts.bedGraph", "_CHG.counts.bedGraph", "_CHH.counts.bedGraph", "_CpG.logit.bedGraph", "_CHG.logit.bedGraph",
            "_CHH.logit.bedGraph", "_CpG.methylKit", "_CHG.methylKit", "_CHH.methylKit", ".cytosine_report.txt"]


def compare_cli(tmp_path, args, env=None, oracle_args=None, ranks=None):
    """same command line through the oracle and the product; same prefix (in different dirs) so headers agree"""
    od, gd = tmp_path / "oracle", tmp_path / "gpu"
    od.mkdir(exist_ok=True), gd.mkdir(exist_ok=True)
    ro = run_oracle(list(oracle_args if oracle_args is not None else args) + ["-o", "out"], cwd=od)
    rg = mdk.run_cli(list(args) + ["-o", "out"], cwd=gd, env=env, ranks=ranks)
    (tmp_path / "gpu_stderr.txt").write_text(rg.stderr)
    if ranks:
        assert all(rc == rg.returncode for rc in rg.rank_returncodes), (rg.rank_returncodes, rg.rank_stderr)
    assert rg.returncode == ro.returncode, (rg.returncode, ro.returncode, rg.stderr[-2000:])
    assert rg.stdout == ro.stdout
    assert [l for l in rg.stderr.splitlines() if l.startswith("loading mappability")] == [l.replace(".bbm", ".bw") for l in ro.stderr.splitlines() if l.startswith("loading mappability")] or oracle_args is None
    seen = 0
    for s in SUFFIXES:
        fo, fg = od / ("out" + s), gd / ("out" + s)
        assert fo.exists() == fg.exists(), s
        if fo.exists():
            seen += 1
            assert filecmp.cmp(fo, fg, shallow=False), f"{s} differs:\n" + diff_head(fo, fg)
    assert seen > 0
    return od, gd


def diff_head(a, b, n=10):
    la, lb = open(a).read().splitlines(), open(b).read().splitlines()
    out = [f"lines: oracle {len(la)} gpu {len(lb)}"]
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            out.append(f"{i}: oracle {x!r} gpu {y!r}")
            if len(out) > n:
                break
    return "\n".join(out)


def G(*names):
    return [str(GOLDEN / n) for n in names]


# the reference's own 15 command lines (tests/test.py) + the same fixtures under more of the option surface
FIXTURE_CMDS = [
    G("ct100.fa", "ct_aln.bam") + ["-q", "2"],
    G("cg100.fa", "cg_aln.bam") + ["-q", "2"],
    G("cg100.fa", "cg_aln.bam") + ["-q", "10"],
    ["--methylKit", "--CHH", "--CHG"] + G("cg100.fa", "cg_aln.bam") + ["-q", "2"],
    ["--minDepth", "2"] + G("cg100.fa", "cg_aln.bam") + ["-q", "2"],
    ["--ignoreFlags", "0xD00"] + G("cg100.fa", "cg_aln.bam") + ["-q", "2"],
    ["--requireFlags", "0xD00"] + G("cg100.fa", "cg_aln.bam") + ["-q", "2"],
    ["--nOT", "50,50,40,40"] + G("cg100.fa", "cg_aln.bam") + ["-q", "2"],
    ["-p", "1", "-q", "0", "--minOppositeDepth", "3", "--maxVariantFrac", "0.25"] + G("cg100.fa", "cg_with_variants.bam"),
    G("chgchh.fa", "chgchh_aln.bam"),
    ["-q", "5"] + G("chgchh.fa", "chgchh_aln.bam"),
    ["-q", "5", "--minConversionEfficiency", "0.9"] + G("chgchh.fa", "chgchh_aln.bam"),
    ["-q", "5", "--minConversionEfficiency", "1.0"] + G("chgchh.fa", "chgchh_aln.bam"),
    ["-q", "1"] + G("cg100.fa", "NH.bam"),
    ["--ignoreNH", "-q", "1"] + G("cg100.fa", "NH.bam"),
    ["--mergeContext", "--CHG", "-q", "2"] + G("cg100.fa", "cg_aln.bam"),
    ["--fraction", "-q", "2", "--CHH"] + G("cg100.fa", "cg_aln.bam"),
    ["--counts", "-q", "2"] + G("cg100.fa", "cg_aln.bam"),
    ["--logit", "-q", "2", "--ignoreFlags", "0"] + G("cg100.fa", "cg_aln.bam"),
    ["--cytosine_report", "--CHG", "--CHH", "-q", "2"] + G("cg100.fa", "cg_aln.bam"),
    ["--mergeContext", "-p", "1", "-q", "0", "--minOppositeDepth", "3", "--maxVariantFrac", "0.25"] + G("cg100.fa", "cg_with_variants.bam"),
    ["-q", "2", "-r", "chrCG:10-50", "--chunkSize", "7"] + G("cg100.fa", "cg_aln.bam"),
    ["-q", "5", "--CHG", "--CHH", "--mergeContext", "--chunkSize", "3"] + G("chgchh.fa", "chgchh_aln.bam"),
]


@pytest.mark.parametrize("args", FIXTURE_CMDS, ids=[" ".join(os.path.basename(a) for a in c) for c in FIXTURE_CMDS])
def test_cli_fixtures_byte_exact(tmp_path, args):
    compare_cli(tmp_path, args)


SYN_CMDS = [
    ("pe", []),
    ("pe", ["--CHG", "--CHH", "--chunkSize", "7000"]),
    ("pe", ["--CHG", "--CHH", "--mergeContext", "--chunkSize", "2500", "-d", "3"]),
    ("pe", ["--chunkSize", "100", "-r", "chrS1:5000-9000", "--CHH"]),
    ("pe", ["-F", "0", "--keepDupes", "--keepSingleton", "--keepDiscordant", "--ignoreNH", "-q", "0", "--chunkSize", "3001", "--CHG"]),
    ("pe", ["--OT", "6,146,6,146", "--OB", "6,146,6,146", "--nOT", "2,3,4,5", "--CHH", "--methylKit"]),
    ("pe", ["-B", "BBM", "--chunkSize", "9999"]),
    ("pe", ["-M", "BW", "-t", "0.6", "-b", "100", "--mergeContext", "--CHG"]),
    ("pe", ["--minOppositeDepth", "2", "--maxVariantFrac", "0.5", "--CHG", "--chunkSize", "8000", "--mergeContext"]),
    ("pe", ["--cytosine_report", "--CHG", "--CHH", "--chunkSize", "6000", "--minOppositeDepth", "2", "--maxVariantFrac", "0.3"]),
    ("pe", ["--fraction", "-p", "20", "-@", "4"]),
    ("pe", ["--logit", "--CHH", "--noCpG"]),
    ("bis", ["--CHG", "--CHH"]),
    ("bis", ["--CTOT", "5,100,5,100", "--nCTOB", "3,3,3,3", "--CHH", "--chunkSize", "5000"]),
    ("pe", ["--OT", "0,0,0,140", "--nOB", "0,9,0,0"]),        # literal zeros: the bounds parser must not depend on a stale errno
    ("se", ["--CHG", "--CHH", "--mergeContext"]),
]


@pytest.mark.parametrize("which,extra", SYN_CMDS, ids=[f"{w}:{' '.join(e)}" for w, e in SYN_CMDS])
@pytest.mark.parametrize("env", [{"MDK_TILE": "512"}, {"MDK_TILE": "1024"}, {"MDK_TILE": "1536"}, {"MDK_TILE": "2048"}, {"MDK_NO_QW": "1"}],
                         ids=["tile512", "tile1024", "tile1536", "tile2048", "lane-per-segment"])
def test_cli_synthetic_byte_exact(tmp_path, small_synth, which, extra, env):
    """every command line under four distinct tile geometries (1 to 4 reference positions per thread -- the library clamps
    larger requests to 2048, test_tile_geometry_is_what_was_asked_for; at 512 positions a tile's segment run overflows the 512
    lanes of its workgroup, so the multi-round path runs too).  Command lines that count CHG or CHH run the quarter-wavefront
    kernel; the last column sends them through the lane-per-segment kernel as well (MDK_NO_QW)"""
    args = [str(small_synth / f"{which}.fa"), str(small_synth / f"{which}.bam")]
    if "BW" in extra:       # the oracle has no bigWig reader: it gets the same track as BBM
        compare_cli(tmp_path, args + [str(small_synth / "pe.bw") if e == "BW" else e for e in extra], env=env,
                    oracle_args=args + [{"BW": str(small_synth / "pe.bbm"), "-M": "-B"}.get(e, e) for e in extra])
        return
    extra = [str(small_synth / "pe.bbm") if e == "BBM" else e for e in extra]
    compare_cli(tmp_path, args + extra, env=env)


def abi_sites(args):
    plan = mdk.Plan(args)
    dev = mdk.Device(plan.dev_cfg())
    got = {}
    while True:
        c = plan.next_chunk()
        if c is None:
            break
        if c.skipped:
            continue
        plan.ensure_reference(dev, c.tid)
        dev.submit(0, c.batch)
        s = dev.download(0)
        prev = -1
        for pos, typ, isg, m, u, off, var in mdk.sites_to_rows(s):
            assert c.beg <= pos < c.end and pos > prev, "sites must be ascending and inside the interval"
            prev = pos
            got[(c.tid, pos)] = (typ, isg, m, u, off, var)
    dev.close()
    plan.close()
    return got


def test_abi_sites_equal_oracle_counters(tmp_path, small_synth):
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--CHG", "--CHH", "--minOppositeDepth", "1", "--chunkSize", "9000"]
    dump = tmp_path / "d.tsv"
    assert run_oracle(args + ["-o", tmp_path / "o"], cwd=tmp_path, dump=dump).returncode == 0
    assert abi_sites(args + ["-o", tmp_path / "g"]) == read_dump(dump)


@pytest.fixture(scope="module")
def s1(tmp_path_factory):
    """BASELINE.json configs[1]: synthetic 1 Mb contig, 30x paired-end WGBS"""
    d = tmp_path_factory.mktemp("s1")
    synth(d / "S1", "-L", "1000000", "-c", "30", "-s", "0x5EED0001")
    return d


def test_s1_cpg_byte_exact(tmp_path, s1):
    compare_cli(tmp_path, [str(s1 / "S1.fa"), str(s1 / "S1.bam")])


def test_s1_config3_chg_chh_trim_byte_exact(tmp_path, s1):
    """BASELINE.json configs[2]"""
    compare_cli(tmp_path, [str(s1 / "S1.fa"), str(s1 / "S1.bam"), "--CHG", "--CHH", "--OT", "6,146,6,146", "--OB", "6,146,6,146"])


def test_s1_merge_variant_byte_exact(tmp_path, s1):
    compare_cli(tmp_path, [str(s1 / "S1.fa"), str(s1 / "S1.bam"), "--mergeContext", "--minOppositeDepth", "5", "--maxVariantFrac", "0.2", "--chunkSize", "250000"])


def test_launch_is_idempotent_and_chunking_invariant(tmp_path, s1):
    """size-independent properties: re-launching a resident batch gives identical sites; total calls do not depend on
    the chunk size or the tile size"""
    base = [str(s1 / "S1.fa"), str(s1 / "S1.bam")]
    a = abi_sites(base + ["-o", tmp_path / "a"])
    os.environ["MDK_TILE"] = "512"
    try:
        b = abi_sites(base + ["--chunkSize", "33333", "-o", tmp_path / "b"])
    finally:
        del os.environ["MDK_TILE"]
    assert a == b
    plan = mdk.Plan(base + ["-o", tmp_path / "c"])
    dev = mdk.Device(plan.dev_cfg())
    c = plan.next_chunk()
    plan.ensure_reference(dev, c.tid)
    dev.upload(0, c.batch)
    dev.launch(0)
    r1 = mdk.sites_to_rows(dev.download(0))
    dev.launch(0)
    r2 = mdk.sites_to_rows(dev.download(0))
    assert r1 == r2 and len(r1) > 1000
    dev.close(), plan.close()


def test_empty_and_edge_batches(tmp_path):
    """empty BAM region, reads hanging over both contig ends, zero-length and clipped alignments"""
    synth(tmp_path / "tiny", "-L", "400,150,3000", "-c", "12", "-s", "5", "-l", "100")
    compare_cli(tmp_path, [str(tmp_path / "tiny.fa"), str(tmp_path / "tiny.bam"), "--CHG", "--CHH", "--chunkSize", "64"])
    compare_cli(tmp_path, [str(tmp_path / "tiny.fa"), str(tmp_path / "tiny.bam"), "-r", "chrS2"])


def test_no_cpu_fallback_symbols():
    """the shipped libraries must not link the oracle"""
    out = subprocess.run(["nm", "-D", str(mdk.LIB_EXTRACT)], capture_output=True, text=True).stdout
    assert "extract_main" in out and "oracle" not in out.lower()


def test_sharded_two_ranks_byte_exact(tmp_path, small_synth):
    """the command as two processes (both on this box's single GPU, so the site buffers travel over the ranks' TCP
    connection instead of ncclSend/ncclRecv): rank 0 writes; output == oracle"""
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--CHG", "--mergeContext", "--chunkSize", "3000", "--minOppositeDepth", "2", "--maxVariantFrac", "0.4"]
    od, gd = compare_cli(tmp_path, args, ranks=2, env={"MDK_DEVICE": "0"})
    assert sorted(os.listdir(od)) == sorted(os.listdir(gd))


def test_effective_bases_hook_matches_evaluator(tmp_path, small_synth):
    """md_dev_debug_effective: the (base, quality) every segment base ends up with after trimming and mate-overlap resolution
    -- the part of the reference that rewrites reads in place (common.c:137-208, overlaps.c:54-119) -- against the slow
    Python evaluator, independent of any counting"""
    import numpy as np
    from batch_eval import Payload, resolve
    args = [str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "--OT", "4,140,6,130", "--nOB", "2,3,4,5", "--chunkSize", "20000", "-o", str(tmp_path / "x")]
    plan = mdk.Plan(args)
    cfg = plan.dev_cfg()
    dev = mdk.Device(cfg)
    c = plan.next_chunk()
    plan.ensure_reference(dev, c.tid)
    dev.upload(0, c.batch)
    n = c.batch.n_segs
    lens = np.array([c.batch.seg[i].len for i in range(n)], dtype=np.uint64)
    off = np.zeros(n, dtype=np.uint64); off[1:] = np.cumsum(lens)[:-1]
    total = int(lens.sum())
    ob = np.zeros(total + 1, dtype=np.uint8); oq = np.zeros(total + 1, dtype=np.uint8)
    rc = dev.L.md_dev_debug_effective(dev.h, 0, ob.ctypes.data, oq.ctypes.data, off.ctypes.data)
    assert rc == 0
    pay, checked, partnered = {}, 0, 0
    for i in range(0, n, 7):            # every 7th segment is plenty
        g = c.batch.seg[i]
        key = (g.off4, g.l_qseq, g.sf & 7, bool(g.sf & 8))
        o = pay.setdefault(key, Payload(c.batch, g.off4, g.l_qseq, g.sf & 7, bool(g.sf & 8), cfg))
        m = None
        if g.sf & 32:
            mk = (g.m_off4, g.m_l_qseq, g.msf & 7, bool(g.msf & 8))
            m = pay.setdefault(mk, Payload(c.batch, g.m_off4, g.m_l_qseq, g.msf & 7, bool(g.msf & 8), cfg))
            partnered += 1
        for j in range(g.len):
            b, q = o.bq(g.q0 + j)
            if m is not None:
                mb, mq = m.bq(g.m_q0 + j)
                q = resolve(bool(g.sf & 16), b, q, mb, mq)
            assert (int(ob[int(off[i]) + j]), int(oq[int(off[i]) + j])) == (b, q), (i, j)
            checked += 1
    assert checked > 20000 and partnered > 50
    dev.close(), plan.close()


def test_high_depth_100x_byte_exact(tmp_path):
    """100x coverage (BASELINE configs[4] depth): several rounds of segments per tile"""
    synth(tmp_path / "deep", "-L", "150000", "-c", "100", "-s", "41", "--bbm")
    compare_cli(tmp_path, [str(tmp_path / "deep.fa"), str(tmp_path / "deep.bam"), "--mergeContext", "--CHG", "-B", str(tmp_path / "deep.bbm")])
    compare_cli(tmp_path, [str(tmp_path / "deep.fa"), str(tmp_path / "deep.bam"), "--CHH", "--minOppositeDepth", "10", "--maxVariantFrac", "0.1"], env={"MDK_TILE": "1024"})


def test_tile_geometry_is_what_was_asked_for():
    """the three geometries of the matrix above are really three: md_dev_open rounds a request up to a multiple of 512 and
    clamps it to 2048 positions (4 per thread)"""
    import ctypes as C
    for ask, want in ((512, 512), (1024, 1024), (0, 2048), (2048, 2048), (700, 1024), (4096, 2048)):
        cfg = mdk.md_dev_cfg(); cfg.keepCpG = 1; cfg.minPhred = 5; cfg.tile = ask
        dev = mdk.Device(cfg, device=0)
        assert mdk.lib_hip().md_dev_tile(dev.h) == want, ask
        dev.close()
    # the library default: 2048 for CpG-only runs, 1536 as soon as CHG or CHH are counted
    cfg = mdk.md_dev_cfg(); cfg.keepCpG = 1; cfg.keepCHH = 1; cfg.minPhred = 5
    dev = mdk.Device(cfg, device=0)
    assert mdk.lib_hip().md_dev_tile(dev.h) == 1536
    dev.close()


def test_group_launch_equals_single_launches(tmp_path):
    """md_dev_launch_group: one kernel over several uploaded chunks (k_pileup_multi) gives every chunk exactly the sites its own
    launch gives; device-prepared and host-prepared slots mixed, empty chunk included"""
    synth(tmp_path / "g", "-L", "30000,2000", "-c", "20", "-s", "77", "--extras")
    args = [str(tmp_path / "g.fa"), str(tmp_path / "g.bam"), "--CHG", "--chunkSize", "4000", "--minOppositeDepth", "2", "-o", str(tmp_path / "x")]
    ph, pd = mdk.Plan(args), mdk.Plan(args)
    pd.set_prep(1)
    cfg = ph.dev_cfg(); cfg.n_slots = 8
    dev = mdk.Device(cfg); dev.set_prep(pd.prep_cfg())
    assert dev.L.md_dev_group_max() == 8

    def sl(s):
        return [(s.site[i].pos, s.site[i].nmeth, s.site[i].nunmeth, s.site[i].meta, s.var[i].noff, s.var[i].nvar) for i in range(s.n_sites)]
    want, k, total = [], 0, 0
    while k < 8:
        ch, cd = ph.next_chunk(), pd.next_chunk()
        assert ch is not None
        ph.ensure_reference(dev, ch.tid)
        if k % 2:
            dev.upload(k, ch.batch)
        else:
            dev.upload_raw(k, cd.raw)
        dev.launch(k)
        want.append(sl(dev.download(k)))
        k += 1
    for group in ([0, 1, 2, 3, 4, 5, 6, 7], [7, 3, 5], [2], [6, 0]):
        dev.launch_group(group)
        for j in group:
            got = sl(dev.download(j))
            assert got == want[j], (group, j)
            total += len(got)
    assert total > 2000
    dev.close(); ph.close(); pd.close()
