#!/usr/bin/env python3
"""TEST TOOL (CPU): the product's HOST preparation (csrc/host: chunk schedule, filter_func's admission, strand, the pairing of mates incl.
htslib's buffer eviction, CIGAR -> segments, packing) -- the path a chunk takes when the device hands it back, and MDK_HOST_PREP=1 -- over
random data shapes and random options: the packed batches are evaluated by the slow test-only evaluator (tests/batch_eval.py) and must
equal the oracle's per-column counters exactly.  usage: fuzz_host_prep.py SEED N [WORKDIR]"""
import random, subprocess, sys, tempfile
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
from conftest import read_dump, run_oracle          # noqa: E402
from test_host_logic import host_counts             # noqa: E402
seed, n = int(sys.argv[1]), int(sys.argv[2]); W = Path(sys.argv[3]) if len(sys.argv) > 3 else Path(tempfile.mkdtemp(prefix="mdk_fuzz_hostprep_")); W.mkdir(parents=True, exist_ok=True)
rnd = random.Random(seed)
shape = ["-l", str(rnd.choice([30, 60, 100, 150]))] + [f for f in ("--single", "--bismark", "--split-records", "--extras") if rnd.random() < 0.4] + ["--bbm"]
subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(W / "s"), "-L", rnd.choice(["14000,5000", "20000", "6000,6000,9000"]), "-c", str(rnd.choice([5, 9, 14])), "-s", str(seed)] + shape, check=True, capture_output=True)
names = [l[1:].split()[0] for l in open(W / "s.fa") if l.startswith(">")]
bad = 0
for it in range(n):
    a = []
    def maybe(p, *opt):
        if rnd.random() < p: a.extend(opt)
    maybe(0.3, "-q", str(rnd.choice([0, 1, 10, 30]))); maybe(0.3, "-p", str(rnd.choice([0, 1, 5, 20, 40])))
    maybe(0.3, "--CHG"); maybe(0.3, "--CHH"); maybe(0.1, "--noCpG")
    maybe(0.2, "--keepDupes"); maybe(0.2, "--keepSingleton"); maybe(0.2, "--keepDiscordant"); maybe(0.15, "--ignoreNH")
    maybe(0.15, "-F", str(rnd.choice([0, 256, 1024, 0xF00, 16]))); maybe(0.1, "-R", str(rnd.choice([0, 1, 2, 3, 64])))
    for o in ("--OT", "--OB", "--CTOT", "--CTOB", "--nOT", "--nOB", "--nCTOB", "--nCTOT"):
        maybe(0.1, o, ",".join(str(rnd.choice([0, 2, 5, 10, 50, 140])) for _ in range(4)))
    maybe(0.6, "--chunkSize", str(rnd.choice([50, 333, 1000, 4001, 1000000]))); maybe(0.15, "--minConversionEfficiency", str(rnd.choice([0.5, 0.9, 1.0])))
    maybe(0.15, "-r", rnd.choice([names[0], f"{names[0]}:2000-4000", f"{names[-1]}:1-3000"]))
    if rnd.random() < 0.2: a += ["-B", str(W / "s.bbm")] + (["-b", str(rnd.choice([1, 15, 60]))] if rnd.random() < 0.5 else []) + (["-t", str(rnd.choice([0.01, 0.5, 0.9]))] if rnd.random() < 0.5 else [])
    variant = rnd.random() < 0.2
    if variant: a += ["--minOppositeDepth", str(rnd.choice([1, 2])), "--maxVariantFrac", str(rnd.choice([0.1, 0.5]))]
    args = [str(W / "s.fa"), str(W / "s.bam")] + a
    d = W / f"d{it}"; d.mkdir(exist_ok=True)
    r = run_oracle(args + ["-o", d / "o"], cwd=d, dump=d / "dump.tsv")
    if r.returncode != 0:
        continue                                    # (a command line the reference refuses: tools/round5/fuzz_options.py compares those)
    want = read_dump(d / "dump.tsv")
    if not variant: want = {k: v[:4] + (0, 0) for k, v in want.items() if v[2] + v[3] > 0}
    try:
        got, _ = host_counts(args + ["-o", str(d / "g")])
    except Exception as e:                          # noqa: BLE001
        bad += 1; print(it, " ".join(a), "-> host side raised", repr(e)[:300], flush=True); continue
    if got != want:
        bad += 1
        ks = sorted(set(got) | set(want)); diff = [(k, want.get(k), got.get(k)) for k in ks if want.get(k) != got.get(k)]
        print(it, " ".join(a), "->", len(diff), "columns differ, first:", diff[:3], flush=True)
print(f"seed {seed} (data: {' '.join(shape)}): {n} command lines, {bad} differing")
sys.exit(1 if bad else 0)
