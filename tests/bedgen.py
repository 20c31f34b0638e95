"""Random BED files for the -l/--keepStrand tests: unsorted, nested, overlapping and duplicated regions, optional
columns, comment/track/browser lines, CRLF endings, coordinates hanging over both contig ends."""
import gzip
import random


def random_bed(path, contigs, n, seed, max_len=400, crlf=False, gz=False, dense=False):
    """contigs: [(name, length)]"""
    rng = random.Random(seed)
    lines = ["# a comment", "track name=x description=\"y z\"", "browser position chr1:1-100"]
    for _ in range(n):
        name, L = rng.choice(contigs)
        kind = rng.random()
        if kind < 0.08:                      # long region that swallows later ones
            s = rng.randrange(0, max(1, L - 10)); e = s + rng.randrange(max_len, 8 * max_len)
        elif kind < 0.12:                    # hangs over the contig end / starts below zero
            s = rng.choice([-7, L - 50, L - 1]); e = L + rng.randrange(1, 100)
        else:
            s = rng.randrange(0, L); e = s + rng.randrange(1, max_len if not dense else 3000)
        cols = [name, str(s), str(e)]
        style = rng.random()
        if style < 0.6:
            cols += ["r%d" % rng.randrange(1000), str(rng.randrange(1000)), rng.choice(["+", "-", ".", "+", "-"])]
        elif style < 0.7:
            cols += ["nm"]
        elif style < 0.8:
            cols += ["nm", "0"]
        sep = "\t" if rng.random() < 0.8 else " "
        lines.append(sep.join(cols) + (rng.choice(["", "\t", "\textra"]) if style < 0.6 else ""))
        if rng.random() < 0.05:
            lines.append(lines[-1])          # exact duplicate
    rng.shuffle(lines)
    text = ("\r\n" if crlf else "\n").join(lines) + "\n"
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(text.encode())
    else:
        with open(path, "w", newline="") as f:
            f.write(text)
    return path
