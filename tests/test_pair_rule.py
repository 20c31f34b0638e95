"""The rule k_prep_segs pairs reads by (methyldackel_amd/csrc/mdk_pair_rule.h), on the host: for random sets of admitted reads -- sorted
starts, random ends, flags and a handful of names -- a plain forward sweep over the reads with one table entry per name (what the overlap
callbacks and htslib's pileup buffer do together, overlaps.c:121-147; the host pipeline's pair_reads is a sweep of this shape) must give
every read the mate, and the earlier/later role, that the per-read evaluation of the kernel gives it: the step-by-step machine over the
reads of its name, and for names with exactly two reads the closed form."""
import subprocess
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent

HARNESS = r"""
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "mdk_pair_rule.h"
struct Read { int32_t pos, rend; uint32_t flag; int name; };
static uint64_t rs = 88172645463325252ULL;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
int main() {
    long sets = 0, reads = 0, bad = 0, twos = 0, mated = 0, evictions = 0;
    const uint32_t flags[6] = {0x1, 0x1 | 0x40, 0x1 | 0x80, 0x1 | 0x8, 0x1 | 0x4, 0x0};
    for(int it = 0; it < 300000; it++) {
        const int n = 1 + rnd() % 12, names = 1 + rnd() % 4, contig = rnd() % 2; int32_t pos = (rnd() % 3 == 0) ? -5 + (int32_t)(rnd() % 8) : (int32_t)(rnd() % 50);
        std::vector<Read> R(n);
        for(int i = 0; i < n; i++) { pos += rnd() % 25; R[i].pos = pos; R[i].rend = pos + (rnd() % 6 == 0 ? 0 : (int32_t)(rnd() % 60)); R[i].flag = flags[rnd() % 6]; R[i].name = rnd() % names; }
        // the sweep: every read in file order against its name's entry (live ends + the pending read)
        std::vector<int> mate(n, -1), second(n, 0);
        std::vector<std::vector<int32_t>> live(names); std::vector<int> pending(names, -1);
        int32_t prev_pos = 0; bool first = true;
        for(int i = 0; i < n; i++) {
            const Read &r = R[i]; const bool inserted = first ? (contig > 0 || r.rend > 0) : r.rend > prev_pos;
            if(inserted) {
                std::vector<int32_t> &L = live[r.name]; bool evicted = false; std::vector<int32_t> keep;
                for(int32_t e : L) { if(!first && e < prev_pos) evicted = true; else keep.push_back(e); }
                L = keep; if(evicted) { pending[r.name] = -1; evictions++; }
                if((r.flag & 1) && !(r.flag & 12)) {
                    if(pending[r.name] < 0) pending[r.name] = i;
                    else { const int a = pending[r.name]; mate[a] = i; mate[i] = a; second[i] = 1; pending[r.name] = -1; }
                }
                L.push_back(r.rend);
            }
            prev_pos = r.pos; first = false;
        }
        // the kernel's way: every read on its own, from the reads of its name
        for(int a = 0; a < n; a++) {
            int got = -1; bool sec = false;
            if(mdk_pairs(R[a].flag)) {
                std::vector<int> idx; for(int i = 0; i < n; i++) if(R[i].name == R[a].name) idx.push_back(i);
                if(idx.size() > MDK_MAXLIVE) continue;                 // (the kernel hands such a chunk to the host)
                MdkPairState S; mdk_pair_init(S);
                for(int x : idx) mdk_pair_step(S, contig, (uint32_t)a, x, R[x].flag, R[x].rend, x == 0, x ? R[x - 1].pos : 0);
                got = S.mate; sec = S.second;
                if(idx.size() == 2) {
                    bool s2 = false; const int f = idx[0], s = idx[1];
                    const int g2 = mdk_pair_two(contig, (uint32_t)a, f, s, R[f].flag, R[f].rend, f == 0, f ? R[f - 1].pos : 0, R[s].flag, R[s].rend, R[s - 1].pos, s2);
                    twos++;
                    if(g2 != got || (got >= 0 && s2 != sec)) bad++;
                }
            }
            if(got != mate[a] || (got >= 0 && (int)sec != second[a])) bad++;
            if(got >= 0) mated++;
            reads++;
        }
        sets++;
    }
    printf("%ld %ld %ld %ld %ld %ld\n", sets, reads, bad, twos, mated, evictions);
    return bad != 0;
}
"""


def test_per_read_rule_equals_the_sweep(tmp_path):
    (tmp_path / "h.cpp").write_text(HARNESS)
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", str(REPO / "methyldackel_amd/csrc"), "-o", str(tmp_path / "h"), str(tmp_path / "h.cpp")], check=True)
    r = subprocess.run([str(tmp_path / "h")], capture_output=True, text=True, timeout=300)
    sets, reads, bad, twos, mated, evictions = map(int, r.stdout.split())
    assert r.returncode == 0 and bad == 0, r.stdout
    assert sets == 300000 and reads > 1_000_000 and twos > 100_000 and mated > 100_000 and evictions > 50_000, r.stdout      # the cases that matter occur
