"""CPU-only: the stand-alone `mergeContext` text tool (mergeContext.c) against the oracle's restatement, and against
`extract --mergeContext` of the oracle (which must give the same CpG lines).  No device is involved."""
import subprocess

import pytest

import methyldackel_amd as mdk
from conftest import ORACLE, run_oracle


def run_tool(args, cwd):
    return subprocess.run([str(mdk.CLI), "mergeContext"] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True)


def run_ref(args, cwd):
    return subprocess.run([str(ORACLE), "mergeContext"] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True)


@pytest.fixture(scope="module")
def graphs(tmp_path_factory, small_synth):
    d = tmp_path_factory.mktemp("mc")
    assert run_oracle([small_synth / "pe.fa", small_synth / "pe.bam", "--CHG", "--CHH", "-o", "s"], cwd=d).returncode == 0
    assert run_oracle([small_synth / "pe.fa", small_synth / "pe.bam", "--CHG", "--mergeContext", "-o", "m"], cwd=d).returncode == 0
    return d


@pytest.mark.parametrize("ctx", ["CpG", "CHG", "CHH"])
def test_same_as_oracle(tmp_path, small_synth, graphs, ctx):
    o = run_ref([small_synth / "pe.fa", graphs / f"s_{ctx}.bedGraph"], cwd=tmp_path)
    g = run_tool([small_synth / "pe.fa", graphs / f"s_{ctx}.bedGraph"], cwd=tmp_path)
    assert o.returncode == 0 and g.returncode == 0
    assert g.stdout == o.stdout and len(o.stdout.splitlines()) > 50


@pytest.mark.parametrize("ctx", ["CpG", "CHG"])
def test_same_as_extract_mergecontext(tmp_path, small_synth, graphs, ctx):
    g = run_tool([small_synth / "pe.fa", graphs / f"s_{ctx}.bedGraph", "-o", tmp_path / "out.bg"], cwd=tmp_path)
    assert g.returncode == 0 and g.stdout == ""
    got = (tmp_path / "out.bg").read_text().splitlines()[1:]
    want = (graphs / f"m_{ctx}.bedGraph").read_text().splitlines()[1:]
    assert got == want


def test_mixed_contexts_in_one_file_and_unknown_contig(tmp_path, small_synth, graphs):
    lines = []
    for ctx in ("CpG", "CHG", "CHH"):
        lines += (graphs / f"s_{ctx}.bedGraph").read_text().splitlines()[1:400]
    lines.sort(key=lambda l: (l.split("\t")[0], int(l.split("\t")[1])))
    mixed = tmp_path / "mixed.bg"
    mixed.write_text("track type=\"bedGraph\"\n" + "\n".join(lines) + "\nchrNope\t5\t6\t100\t1\t0\nchrS1\t1\t2\t0\t0\t1\n")
    o, g = run_ref([small_synth / "pe.fa", mixed], cwd=tmp_path), run_tool([small_synth / "pe.fa", mixed], cwd=tmp_path)
    assert o.returncode == 0 and g.returncode == 0
    assert g.stdout == o.stdout
    assert g.stderr == o.stderr and "unknown chromosome name" in g.stderr


BAD = [([], 0), (["x.fa"], 255), (["nope.fa", "x.bg"], 254), (["FA", "nope.bg"], 253), (["FA", "x.bg", "-o", "/nonexistent/d/x"], 2), (["--bogus", "FA", "x.bg"], 1)]


@pytest.mark.parametrize("args,rc", BAD, ids=[" ".join(b[0]) or "no arguments" for b in BAD])
def test_option_errors(tmp_path, small_synth, args, rc):
    args = [str(small_synth / "pe.fa") if a == "FA" else a for a in args]
    o, g = run_ref(args, cwd=tmp_path), run_tool(args, cwd=tmp_path)
    assert o.returncode == rc and g.returncode == rc

    def message(s):
        out = []
        for l in s.splitlines():
            if l.startswith("Usage:"):
                break
            if l.strip() and "invalid option" not in l and "unrecognized option" not in l:
                out.append(l)
        return out
    assert message(g.stderr) == message(o.stderr)
