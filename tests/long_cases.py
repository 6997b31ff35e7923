"""Long reads built to exercise the long-read branch of the extension stage (SURVEY.md App. A.6 without MM_F_SR: oracle/mm_align.c
align1_lr, oracle/mm_rmq.c): plain noisy reads, chimeras, large deletions / insertions (z-drop splits, long joins), inverted segments
(the inversion test and mm_align1_inv), reads near the identity at which min_dp_max bites, clipped ends, N runs, tandem duplications
(overlapping chains, RMQ re-chain).  Shared by the CPU and GPU suites."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", np.uint8)
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _noisy(rng, src, sub, indel):
    out = bytearray()
    for c in bytes(src):
        u = rng.random()
        if u < indel:
            continue
        if u < 2 * indel:
            out.append(ACGT[rng.integers(0, 4)])
        out.append(ACGT[rng.integers(0, 4)] if u > 1.0 - sub else c)
    return bytes(out)


def _rc(b):
    return bytes(b).translate(COMP)[::-1]


def long_edge_reads(ref, n=240, seed=11, max_len=6000):
    rng = np.random.default_rng(seed)
    G = len(ref)

    def rnd(m):
        return bytes(ACGT[rng.integers(0, 4, m)])

    def locus(L):
        s = int(rng.integers(0, G - L - 1))
        return s, bytes(ref[s:s + L])
    recs = []
    for it in range(n):
        kind = it % 13
        L = int(min(max(rng.lognormal(7.6, 0.5), 700), max_len))
        if kind == 0:
            r = _noisy(rng, locus(L)[1], 0.02, 0.015)
        elif kind == 1:      # chimera of two loci, the second on the other strand half of the time
            a, b = locus(L // 2)[1], locus(L // 2)[1]
            r = _noisy(rng, a, 0.02, 0.015) + _noisy(rng, _rc(b) if rng.random() < 0.5 else b, 0.02, 0.015)
        elif kind == 2:      # a stretch of the reference missing from the read
            s, src = locus(L + 2500)
            d = int(rng.integers(300, 2000)); p = int(rng.integers(300, L - 300))
            r = _noisy(rng, src[:p] + src[p + d:p + d + (L - p)], 0.02, 0.015)
        elif kind == 3:      # a stretch of random sequence inserted into the read
            s, src = locus(L)
            d = int(rng.integers(300, 1500)); p = int(rng.integers(300, L - 300))
            r = _noisy(rng, src[:p], 0.02, 0.015) + rnd(d) + _noisy(rng, src[p:], 0.02, 0.015)
        elif kind == 4:      # an inverted segment in the middle
            s, src = locus(L)
            d = int(rng.integers(250, 900)); p = int(rng.integers(300, max(301, L - 300 - d)))
            r = _noisy(rng, src[:p] + _rc(src[p:p + d]) + src[p + d:], 0.015, 0.01)
        elif kind == 5:      # divergent reads: chains exist, regions may fall below min_dp_max / min_chain_score
            e = float(rng.uniform(0.10, 0.24))
            r = _noisy(rng, locus(L)[1], e * 0.5, e * 0.25)
        elif kind == 6:      # short and noisy
            L2 = int(rng.integers(160, 420)); e = float(rng.uniform(0.06, 0.16))
            r = _noisy(rng, locus(L2)[1], e * 0.5, e * 0.25)
        elif kind == 7:      # clipped ends
            r = rnd(int(rng.integers(100, 600))) + _noisy(rng, locus(L)[1], 0.02, 0.015) + rnd(int(rng.integers(100, 800)))
        elif kind == 8:      # a run of Ns
            src = bytearray(locus(L)[1]); p = int(rng.integers(200, L - 300)); m = int(rng.integers(5, 120))
            src[p:p + m] = b"N" * m
            r = _noisy(rng, bytes(src), 0.02, 0.015)
        elif kind == 9:      # tandem duplication of a segment inside the read
            s, src = locus(L)
            d = int(rng.integers(200, 900)); p = int(rng.integers(300, max(301, L - 300 - d)))
            r = _noisy(rng, src[:p + d] + src[p:], 0.02, 0.015)
        elif kind == 10:     # unrelated sequence
            r = rnd(L)
        elif kind == 12:     # a few exact k-mers on one diagonal with unrelated sequence between them: a chain, hardly an alignment
            m = int(rng.integers(3, 6)); s, src = locus(600)
            parts, pos = [rnd(int(rng.integers(20, 200)))], 0
            for _ in range(m):
                kl = int(rng.integers(15, 20)); g = int(rng.integers(5, 45))
                parts += [src[pos:pos + kl], rnd(g)]; pos += kl + g
            r = b"".join(parts) + rnd(int(rng.integers(20, 200)))
        else:                # accurate read (HiFi-like)
            r = _noisy(rng, locus(L)[1], 0.002, 0.001)
        if it % 3 == 1:
            r = _rc(r)
        recs.append(r)
    bases = np.frombuffer(b"".join(recs), np.uint8)
    offs = np.zeros(len(recs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(x) for x in recs])
    return recs, bases, offs


def tandem_case(seed=5, n_arrays=24, n_reads=240):
    """A reference with short PERFECT tandem arrays (4-8 copies of a 40-180 bp monomer, below any occurrence cut-off) in unique sequence,
    and reads across them with a different copy number than the reference.  The anchors of such a read form a lattice - (x + m P, y + n P)
    for every pair of copies - and two lattice points on one anti-diagonal reached by mirror-image gaps carry the same chaining score: the
    long join's range-minimum query meets candidates of EQUAL priority, which krmq_rmq resolves by the shape of its tree
    (oracle/mm_rmq.c; scrubby_amd/csrc/sh_rmq_tree.h).  Returns (contigs, bases, offsets)."""
    rng = np.random.default_rng(seed)

    def rnd(m):
        return bytes(ACGT[rng.integers(0, 4, m)])
    parts, arrays, pos = [], [], 0
    for _ in range(n_arrays):
        flank = rnd(int(rng.integers(6000, 9000)))
        mono = rnd(int(rng.integers(40, 180)))
        copies = int(rng.integers(4, 9))
        parts += [flank, mono * copies]
        arrays.append((pos + len(flank), mono, copies))
        pos += len(flank) + len(mono) * copies
    parts.append(rnd(8000))
    ref = b"".join(parts)
    recs = []
    for it in range(n_reads):
        st, mono, copies = arrays[it % n_arrays]
        left = int(rng.integers(600, 2500)); right = int(rng.integers(600, 2500))
        c2 = max(2, copies + int(rng.integers(-2, 3)))                       # the read's copy number
        src = ref[st - left:st] + mono * c2 + ref[st + len(mono) * copies:st + len(mono) * copies + right]
        e = (0.0, 0.004, 0.02)[it % 3]                                       # exact, HiFi-like, ONT-like
        r = _noisy(rng, src, e, e * 0.75) if e > 0 else src
        if it % 4 == 1:
            r = _rc(r)
        recs.append(r)
    bases = np.frombuffer(b"".join(recs), np.uint8)
    offs = np.zeros(len(recs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(x) for x in recs])
    return [ref], bases, offs


def dense_lattice_case(seed=11, n_reads=6, ref_copies=10, read_copies=(30, 38)):
    """Reads whose long join holds more anchors inside rmq_inner_dist (1000 reference bases) than the 4096-anchor LDS ring takes: a perfect
    tandem array of ref_copies copies of a ~100-bp monomer (every k-mer at most ref_copies times in the reference: below map-ont's
    min_mid_occ, nothing is filtered) read by a molecule with 30 - 38 copies between long unique flanks (so that a k-mer's copies stay below
    q_occ_frac of the read's minimizers and mm_seed_mz_flt keeps them) - ref_copies x read_copies lattice points per minimizer of the
    monomer, ~5 000 - 7 000 anchors over one kilobase of reference.  Returns (contigs, bases, offsets)."""
    rng = np.random.default_rng(seed)

    def rnd(m):
        return bytes(ACGT[rng.integers(0, 4, m)])
    left, right = rnd(9000), rnd(9000)
    mono = rnd(int(rng.integers(96, 110)))
    ref = left + mono * ref_copies + right
    recs = []
    for it in range(n_reads):
        c2 = int(rng.integers(read_copies[0], read_copies[1] + 1))
        src = left[-int(rng.integers(8000, 9000)):] + mono * c2 + right[:int(rng.integers(8000, 9000))]
        e = (0.0, 0.004)[it % 2]
        r = _noisy(rng, src, e, e * 0.75) if e > 0 else src
        if it % 3 == 2:
            r = _rc(r)
        recs.append(r)
    bases = np.frombuffer(b"".join(recs), np.uint8)
    offs = np.zeros(len(recs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(x) for x in recs])
    return [ref], bases, offs
