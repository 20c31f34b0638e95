#!/usr/bin/env python3
"""TEST TOOL (CPU): the one-process-per-GPU driver of the command (csrc/host/mdk_ranks.c) on the device stand-in, N = 2..4 ranks on this
host: random chunk sizes, options, dealt or claimed chunks, with or without the index, chunks handed back to the host -- every output file
against the oracle's, every rank's exit code.  usage: fuzz_ranks.py SEED N [WORKDIR]"""
import os, random, shutil, socket, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
seed, n = int(sys.argv[1]), int(sys.argv[2]); W = Path(sys.argv[3] if len(sys.argv) > 3 else f"/tmp/mdk_fuzz_ranks_{seed}"); W.mkdir(parents=True, exist_ok=True)
rnd = random.Random(seed)
# MDK_FUZZ_CLI: another build of the command (a sanitizer build with the stand-in linked in: then nothing is preloaded)
CLI = os.environ.get("MDK_FUZZ_CLI", str(REPO / "methyldackel_amd/_build/MethylDackel"))
if not (W / "s.bam").exists():
    subprocess.run([str(REPO / "tools/_build/mdk_synth"), "-o", str(W / "s"), "-L", rnd.choice(["150000,60000", "90000,30000,30000", "200000"]), "-c", str(rnd.choice([8, 16])), "-s", str(seed), "--extras"] + (["--split-records"] if rnd.random() < 0.4 else []), check=True, capture_output=True)
bad = 0
for it in range(n):
    a = []
    if rnd.random() < 0.4: a.append("--CHG")
    if rnd.random() < 0.3: a.append("--CHH")
    if rnd.random() < 0.3: a.append("--mergeContext")
    if rnd.random() < 0.2: a += rnd.choice([["--counts"], ["--fraction"], ["--methylKit"], ["--cytosine_report"]])
    if "--methylKit" in a or "--cytosine_report" in a: a = [x for x in a if x != "--mergeContext"]
    if rnd.random() < 0.2: a += ["--minOppositeDepth", "2", "--maxVariantFrac", "0.3"]
    a += ["--chunkSize", str(rnd.choice([1500, 4000, 9000, 20000, 70000, 1000000]))]
    world = rnd.choice([2, 2, 3, 4])
    env = {**({} if os.environ.get("MDK_FUZZ_CLI") else {"LD_PRELOAD": str(REPO / "tools/_build/libmdk_dev_standin.so")}), "MDK_STANDIN_DUMP": str(W / "dump.tsv"), "HSA_DISABLE_COREDUMP_ON_EXCEPTION": "1"}
    if rnd.random() < 0.4: env["MDK_CLAIM"] = "1"
    if rnd.random() < 0.3: env["MDK_NO_INDEX"] = "1"
    if rnd.random() < 0.3: env["MDK_STANDIN_HANDBACK"] = str(rnd.choice([2, 3, 5]))
    if rnd.random() < 0.3: env["MDK_STANDIN_US_PER_KREC"] = str(rnd.choice([2000, 20000]))
    if rnd.random() < 0.2: env["MDK_GROUPS_IN_FLIGHT"] = str(rnd.choice([2, 5]))
    od, gd = W / "o", W / "g"
    for d in (od, gd): shutil.rmtree(d, ignore_errors=True); d.mkdir()
    args = a + [str(W / "s.fa"), str(W / "s.bam"), "-@", "3", "-o", "out"]
    o = subprocess.run([str(REPO / "oracle/_build/mdk_oracle"), "extract"] + args, cwd=od, env=dict(os.environ, MDK_ORACLE_DUMP=str(W / "dump.tsv")), capture_output=True, text=True, timeout=300)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(world):
        e = dict(os.environ); e.update(env); e.update({"MDK_WORLD": str(world), "MDK_RANK": str(r), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        procs.append(subprocess.Popen([CLI, "extract"] + args, cwd=gd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    why = []
    try:
        outs = [p.communicate(timeout=300) for p in procs]
        rcs = [p.returncode for p in procs]
        if any(rc != o.returncode for rc in rcs): why.append(f"exit codes {rcs} (oracle {o.returncode}) {outs[0][1][-300:]}")
    except subprocess.TimeoutExpired:
        for p in procs: p.kill()
        why.append("HANG")
    fo, fg = sorted(p.name for p in od.iterdir()), sorted(p.name for p in gd.iterdir())
    if fo != fg: why.append(f"files {fo} vs {fg}")
    else:
        for f in fo:
            if (od / f).read_bytes() != (gd / f).read_bytes(): why.append(f"{f} differs")
    if why:
        bad += 1; print(it, world, " ".join(a), {k: v for k, v in env.items() if k.startswith("MDK_") and k != "MDK_STANDIN_DUMP"}, "->", "; ".join(why)[:700], flush=True)
print(f"seed {seed}: {n} runs of 2-4 ranks, {bad} differing")
sys.exit(1 if bad else 0)
