"""Differential search for reference vector t8 (reference tests/test.py:82-88: `--nOT 50,50,40,40` on cg_aln.bam asserts
12 lines; the code of common.c:174-208 + overlaps.c:54-119, this oracle and the product give 11).

Runs all 15 reference vectors under every single-rule deviation the oracle knows (MDK_ORACLE_PERTURB=n, oracle/mdk_oracle.c)
and prints which deviations reproduce 12 lines on t8 while keeping the other 14.  Not a test: a diagnostic whose table is
recorded in DESIGN.md section 3.  Usage: python tests/t8_differential.py"""
import os
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from test_oracle_reference_vectors import CASES, resolve, count_lines  # noqa: E402

ORACLE = HERE.parent / "oracle/_build/mdk_oracle"
T8 = ("t8:82-88", ["--nOT", "50,50,40,40", "cg100.fa", "cg_aln.bam", "-q", "2"], {"_CpG.bedGraph": 12})
NAMES = ["reference code (no deviation)", "abs right trim masks rb-1 bases", "abs left trim masks lb-1 bases", "abs right trim masks rb+1 bases",
         "abs left trim masks lb+1 bases", "read #1 uses the read #2 pair of bounds and vice versa", "--nOT read as --OT (keep [lb,rb))",
         "abs bounds counted from the 5' end of the sequenced read (swapped for reverse-strand records)", "abs trim zeroes quality only (base kept)",
         "abs trim sets N only (quality kept)", "paired reads: abs trim after the overlap rule", "overlap rule: 'a' is the later record",
         "overlap rule: equal quality favours the first record", "overlap rule skips positions where either base is N", "no overlap rule at all",
         "abs trim not applied to read #1", "abs trim not applied to read #2", "QC-fail records admitted"]


def run(pt):
    out = {}
    for name, args, expect in CASES + [T8]:
        with tempfile.TemporaryDirectory() as td:
            env = dict(os.environ, MDK_ORACLE_PERTURB=str(pt))
            r = subprocess.run([str(ORACLE), "extract"] + [str(a) for a in resolve(args)] + ["-o", td + "/t"], cwd=td, env=env, capture_output=True, text=True)
            ok = r.returncode == 0
            got = {}
            for suffix, want in expect.items():
                n = count_lines(td + "/t" + suffix) if ok and os.path.exists(td + "/t" + suffix) else -1
                got[suffix] = n
                ok = ok and (n > 1 if want == ">1" else n == want)
            out[name.split(":")[0]] = (ok, got)
    return out


if __name__ == "__main__":
    print("| # | single deviation | t8 lines | other 14 vectors | broken |\n|---|---|---|---|---|")
    for pt, nm in enumerate(NAMES):
        res = run(pt)
        t8 = res["t8"][1]["_CpG.bedGraph"]
        broken = [k for k, (ok, _) in res.items() if k != "t8" and not ok]
        print(f"| {pt} | {nm} | {t8} | {'kept' if not broken else 'BROKEN'} | {' '.join(broken)} |")
