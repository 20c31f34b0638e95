#!/usr/bin/env python3
"""Regenerates tests/golden/expected/: the output of THIS REPO'S ORACLE (oracle/mdk_oracle) for a fixed list of command
lines over the reference's fixture BAMs.  These are NOT outputs of the reference program (it cannot be built here); they
pin the oracle against accidental drift between rounds, and give the product a set of files to be compared with that
does not depend on rebuilding the oracle.  Usage: python tests/golden/make_expected.py   (from the repo root)"""
import pathlib
import subprocess
import sys
import tempfile

HERE = pathlib.Path(__file__).resolve().parent
ORACLE = HERE.parent.parent / "oracle" / "_build" / "mdk_oracle"

# name -> (command, arguments); fixture files are given by name and resolved against tests/golden/
COMMANDS = {
    "extract_cg_q2": ("extract", ["cg100.fa", "cg_aln.bam", "-q", "2"]),
    "extract_cg_all_contexts": ("extract", ["cg100.fa", "cg_aln.bam", "-q", "2", "--CHG", "--CHH"]),
    "extract_cg_methylkit": ("extract", ["--methylKit", "--CHH", "--CHG", "cg100.fa", "cg_aln.bam", "-q", "2"]),
    "extract_cg_merge": ("extract", ["--mergeContext", "--CHG", "cg100.fa", "cg_aln.bam", "-q", "2"]),
    "extract_cg_cytosine_report": ("extract", ["--cytosine_report", "--CHG", "--CHH", "cg100.fa", "cg_aln.bam", "-q", "2"]),
    "extract_cg_fraction": ("extract", ["--fraction", "cg100.fa", "cg_aln.bam", "-q", "2"]),
    "extract_cg_logit": ("extract", ["--logit", "cg100.fa", "cg_aln.bam", "-q", "2", "--ignoreFlags", "0"]),
    "extract_cg_trim": ("extract", ["--nOT", "50,50,40,40", "cg100.fa", "cg_aln.bam", "-q", "2"]),
    "extract_variants": ("extract", ["-p", "1", "-q", "0", "--minOppositeDepth", "3", "--maxVariantFrac", "0.25", "cg100.fa", "cg_with_variants.bam"]),
    "extract_chgchh": ("extract", ["-q", "5", "--CHG", "--CHH", "chgchh.fa", "chgchh_aln.bam"]),
    "extract_chgchh_conv": ("extract", ["-q", "5", "--minConversionEfficiency", "0.9", "chgchh.fa", "chgchh_aln.bam"]),
    "extract_ct": ("extract", ["ct100.fa", "ct_aln.bam", "-q", "2", "--CHH"]),
    "extract_nh": ("extract", ["--ignoreNH", "-q", "1", "cg100.fa", "NH.bam"]),
    "mbias_cg": ("mbias", ["cg100.fa", "cg_aln.bam", "out", "--txt", "-q", "2"]),
    "mbias_chgchh": ("mbias", ["chgchh.fa", "chgchh_aln.bam", "out", "--txt", "-q", "5", "--CHG", "--CHH"]),
    "perread_cg": ("perRead", ["cg100.fa", "cg_aln.bam", "-q", "2", "-o", "out.perRead.txt"]),
    "perread_chgchh": ("perRead", ["chgchh.fa", "chgchh_aln.bam", "-q", "5", "-p", "20", "-o", "out.perRead.txt"]),
}
FIXTURES = {p.name for p in HERE.iterdir() if p.suffix in (".fa", ".bam")}


def run(name, tool=ORACLE, prefix_args=()):
    """-> {relative file name: bytes} of everything the command wrote (+ stdout), run in an empty directory with prefix `out`"""
    cmd, args = COMMANDS[name]
    args = [str(HERE / a) if a in FIXTURES else a for a in args]
    if cmd == "extract":
        args += ["-o", "out"]
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([str(tool)] + list(prefix_args) + [cmd] + args, cwd=d, capture_output=True)
        assert r.returncode == 0, (name, r.stderr.decode()[-500:])
        out = {p.name: p.read_bytes() for p in sorted(pathlib.Path(d).iterdir())}
        out["stdout"] = r.stdout
        if cmd == "mbias":
            out["suggestion"] = b"".join(l + b"\n" for l in r.stderr.splitlines() if l.startswith(b"Suggested"))
        return out


if __name__ == "__main__":
    dst = HERE / "expected"
    dst.mkdir(exist_ok=True)
    for old in dst.iterdir():
        old.unlink()
    n = 0
    for name in COMMANDS:
        for fn, data in run(name).items():
            (dst / f"{name}.{fn}").write_bytes(data); n += 1
    print(f"wrote {n} files to {dst}", file=sys.stderr)
