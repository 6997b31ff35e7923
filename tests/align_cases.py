"""Reads built to exercise the extension stage (App. A.6): chains that exist but whose regions mm_filter_regs may drop, z-drops in the
ungapped middle, indels next to the anchors, Ns, short reads, both strands.  Shared by the CPU and GPU suites."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", np.uint8)


def edge_reads(ref, n=4000, seed=5):
    rng = np.random.default_rng(seed)

    def rnd(m):
        return ACGT[rng.integers(0, 4, m)]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    recs = []
    for it in range(n):
        s = int(rng.integers(1000, len(ref) - 2000))
        kind = it % 10
        if kind == 0:      # two exact ends, garbage middle: z-drop in the ungapped stretch, mm_split_reg
            a, c, L = int(rng.integers(21, 40)), int(rng.integers(21, 40)), 150
            r = ref[s:s + L].copy(); r[a:L - c] = rnd(L - c - a)
        elif kind == 1:    # one short exact core in garbage
            a = int(rng.integers(22, 34)); r = rnd(150); p = int(rng.integers(0, 150 - a)); r[p:p + a] = ref[s:s + a]
        elif kind == 2:    # two cores on different diagonals: single-anchor stretches
            a, g, d = int(rng.integers(21, 30)), int(rng.integers(1, 30)), int(rng.integers(-5, 6))
            r = np.concatenate([rnd(20), ref[s:s + a], rnd(g), ref[s + a + g + d:s + a + g + d + a], rnd(150)])[:150]
        elif kind == 3:    # heavy substitutions
            r = ref[s:s + 150].copy(); m = rng.random(150) < rng.uniform(0.05, 0.25); r[m] = rnd(int(m.sum()))
        elif kind == 4:    # indels
            r = list(ref[s:s + 170])
            for _ in range(int(rng.integers(1, 6))):
                p = int(rng.integers(5, len(r) - 5))
                if rng.random() < 0.5:
                    del r[p:p + int(rng.integers(1, 4))]
                else:
                    r[p:p] = list(rnd(int(rng.integers(1, 4))))
            r = np.array(r[:150], np.uint8)
        elif kind == 5:    # Ns
            r = ref[s:s + 150].copy(); r[rng.integers(0, 150, int(rng.integers(1, 12)))] = ord("N")
        elif kind == 6:    # short reads
            L = int(rng.integers(25, 60)); r = ref[s:s + L].copy()
            if rng.random() < 0.5:
                r[int(rng.integers(0, L))] = ACGT[int(rng.integers(0, 4))]
        elif kind == 7:    # reverse strand with errors
            r = np.frombuffer(bytes(ref[s:s + 150]).translate(comp)[::-1], np.uint8).copy(); m = rng.random(150) < 0.06; r[m] = rnd(int(m.sum()))
        elif kind == 8:    # indel in a homopolymer right behind the last anchor: the gap is left-aligned into the stretch
            r = ref[s:s + 150].copy(); p = int(rng.integers(100, 135)); run = int(rng.integers(4, 12))
            r[p:p + run] = r[p]; r = np.concatenate([r[:p + run], r[p:p + 1], r[p + run:]])[:150]
        else:              # 250-bp reads with a few errors (longer extensions)
            r = ref[s:s + 250].copy(); m = rng.random(250) < 0.03; r[m] = rnd(int(m.sum()))
        recs.append(bytes(r))
    bases = np.frombuffer(b"".join(recs), np.uint8)
    offs = np.zeros(len(recs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(x) for x in recs])
    return recs, bases, offs
