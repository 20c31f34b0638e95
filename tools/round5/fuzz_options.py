#!/usr/bin/env python3
"""TEST TOOL (CPU): random `extract` command lines through the oracle and through the command on the device stand-in (tools/dev_standin.c:
the counting is the oracle's, everything around it -- option parsing, chunk schedule, the host's variant filter, context merging, the
emitters and their formats, messages and exit codes -- the product's): exit codes, every output file byte for byte, and stderr where the
command refuses its options.  usage: fuzz_options.py SEED N [WORKDIR]"""
import os, random, shutil, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
seed, n = int(sys.argv[1]), int(sys.argv[2]); W = Path(sys.argv[3] if len(sys.argv) > 3 else "/tmp/mdk_fuzz_options"); W.mkdir(parents=True, exist_ok=True)
rnd = random.Random(seed)
# MDK_FUZZ_CLI: another build of the command (a sanitizer build with the stand-in linked in: then nothing is preloaded)
CLI = os.environ.get("MDK_FUZZ_CLI", str(REPO / "methyldackel_amd/_build/MethylDackel"))
# the data set follows from the seed too: read length, single or paired ends, Bismark-style records, records split across BGZF members,
# with or without an index, with mappability tracks
shape = ["-l", str(rnd.choice([40, 75, 100, 150, 250]))] + [f for f in ("--single", "--bismark", "--split-records", "--no-bai", "--extras") if rnd.random() < 0.35] + ["--bbm", "--bw"]
if not (W / "s.bam").exists():
    subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(W / "s"), "-L", rnd.choice(["90000,30000", "120000", "20000,20000,50000"]), "-c", str(rnd.choice([6, 14, 30])), "-s", str(seed)] + shape, check=True, capture_output=True)
names = [l[1:].split()[0] for l in open(W / "s.fa") if l.startswith(">")]
(W / "r.bed").write_text(f"{names[0]}\t1000\t30000\n{names[0]}\t45000\t46000\tx\t0\t-\n{names[-1]}\t0\t9000\tx\t0\t+\n")
def pick():
    a = []
    def maybe(p, *opt):
        if rnd.random() < p: a.extend(opt)
    maybe(0.25, "-q", str(rnd.choice([0, 1, 10, 30, 60]))); maybe(0.25, "-p", str(rnd.choice([0, 1, 5, 20, 40])))
    maybe(0.2, "-d", str(rnd.choice([0, 1, 2, 5, 10]))); maybe(0.12, "--fraction"); maybe(0.12, "--counts"); maybe(0.1, "--logit"); maybe(0.12, "--methylKit"); maybe(0.12, "--cytosine_report")
    maybe(0.15, "--noCpG"); maybe(0.3, "--CHG"); maybe(0.3, "--CHH"); maybe(0.25, "--mergeContext"); maybe(0.1, "--keepStrand"); maybe(0.15, "-l", str(W / "r.bed"))
    maybe(0.15, "--keepDupes"); maybe(0.15, "--keepSingleton"); maybe(0.15, "--keepDiscordant"); maybe(0.1, "--ignoreNH")
    maybe(0.1, "-F", str(rnd.choice([0, 256, 1024, 0xF00, 3840, 16]))); maybe(0.1, "-R", str(rnd.choice([0, 1, 2, 3, 64])))
    for o in ("--OT", "--OB", "--CTOT", "--CTOB", "--nOT", "--nOB", "--nCTOT", "--nCTOB"):
        maybe(0.08, o, ",".join(str(rnd.choice([0, 2, 5, 10, 140, 148])) for _ in range(4)))
    maybe(0.15, "--minOppositeDepth", str(rnd.choice([0, 1, 3]))); maybe(0.15, "--maxVariantFrac", str(rnd.choice([0.0, 0.1, 0.5, 1.0])))
    maybe(0.5, "--chunkSize", str(rnd.choice([1, 100, 777, 5000, 33333, 1000000]))); maybe(0.1, "--minConversionEfficiency", str(rnd.choice([0.0, 0.5, 0.9, 1.0])))
    maybe(0.15, "-r", rnd.choice([names[0], f"{names[0]}:2000-40000", f"{names[-1]}:1-5000", f"{names[0]}:50,000", "nosuchcontig", f"{names[0]}:70000-60000"]))
    maybe(0.3, "-@", str(rnd.choice([1, 2, 4, 7])))
    if rnd.random() < 0.2:
        a.extend(rnd.choice([["-M", str(W / "s.bw")], ["-B", str(W / "s.bbm")]]))
        maybe(0.5, "-t", str(rnd.choice([0.01, 0.5, 0.9]))); maybe(0.5, "-b", str(rnd.choice([1, 15, 60])))
    rnd.shuffle(a) if rnd.random() < 0.0 else None
    return a
env_std = dict(os.environ)
bad = refused = 0
for it in range(n):
    a = pick()
    od, gd = W / "o", W / "g"
    for d in (od, gd): shutil.rmtree(d, ignore_errors=True); d.mkdir()
    args = a + [str(W / "s.fa"), str(W / "s.bam"), "-o", "out"]
    eo = dict(env_std, MDK_ORACLE_DUMP=str(W / "dump.tsv"))
    # (the oracle reads mappability from a BBM file only -- no libBigWig here --; mdk_synth wrote the same track both ways)
    oargs = [str(W / "s.bbm") if x == str(W / "s.bw") else "-B" if x == "-M" else x for x in args]
    o = subprocess.run([str(REPO / "oracle/_build/mdk_oracle"), "extract"] + oargs, cwd=od, env=eo, capture_output=True, text=True, timeout=300)
    eg = dict(env_std, **({} if os.environ.get("MDK_FUZZ_CLI") else {"LD_PRELOAD": str(REPO / "tools/_build/libmdk_dev_standin.so")}), MDK_STANDIN_DUMP=str(W / "dump.tsv"), HSA_DISABLE_COREDUMP_ON_EXCEPTION="1")
    try:
        g = subprocess.run([CLI, "extract"] + args, cwd=gd, env=eg, capture_output=True, text=True, timeout=300); grc = g.returncode; gerr = g.stderr
    except subprocess.TimeoutExpired:
        grc, gerr = "HANG", ""
    why = []
    refused += o.returncode != 0
    if grc != o.returncode: why.append(f"exit {o.returncode} vs {grc}")
    fo, fg = sorted(p.name for p in od.iterdir()), sorted(p.name for p in gd.iterdir())
    if fo != fg: why.append(f"files {fo} vs {fg}")
    else:
        for f in fo:
            if (od / f).read_bytes() != (gd / f).read_bytes(): why.append(f"{f} differs")
    # (the oracle abbreviates the usage text: what is compared is the first line either side says)
    gerr = gerr.replace(str(W / "s.bw"), str(W / "s.bbm"))
    if o.returncode != 0 and isinstance(grc, int) and o.stderr.strip().splitlines()[:1] != [l for l in gerr.strip().splitlines() if not l.startswith("[mdk")][:1]:
        why.append(f"message: {o.stderr.strip().splitlines()[:1]} vs {gerr.strip().splitlines()[:1]}")
    if why:
        bad += 1; print(it, " ".join(a), "->", "; ".join(why)[:600], flush=True)
print(f"seed {seed} (data: {' '.join(shape)}): {n} command lines ({refused} of them refused by the oracle), {bad} differing")
sys.exit(1 if bad else 0)
