"""Seeded synthetic RNA-seq read generator (host side, numpy) — SURVEY.md §8(d).

transcriptome: G bases i.i.d. uniform over ACGT, cut into transcripts of length U[500,4000];
expression: log-normal(sigma) weights (or uniform); fragments: N(300,30) clipped to [L, transcript];
reads: both fragment ends, RIGHT read emitted as sequenced (reverse complement of the fragment's
right end, i.e. the library needs `-revcomp-right`); substitution error rate `err` per base
(error bases get quality '#', PHRED 2 < default minimum 3, everything else 'I'); `n_rate` of bases
replaced by 'N'.  The same arrays feed the CPU oracle and the HIP path.
"""
import numpy as np

_COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[a] = b


def revcomp(a):
    return _COMP[a[..., ::-1]]


def make_transcriptome(G, seed=0x5EED, tmin=500, tmax=4000):
    rng = np.random.default_rng(seed)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, G)]
    bounds = [0]
    while bounds[-1] < G:
        bounds.append(min(G, bounds[-1] + int(rng.integers(tmin, tmax + 1))))
    b = np.asarray(bounds, np.int64)
    if b[-1] - b[-2] < tmin and len(b) > 2:          # merge a short tail into its neighbour
        b = np.delete(b, -2)
    return genome, b[:-1], b[1:] - b[:-1]


def generate_pairs(n_pairs, G=1 << 20, L=150, err=0.001, n_rate=1e-4, sigma=2.0, seed=0x5EED,
                   frag_mean=300.0, frag_sd=30.0, uniform_expr=False):
    """returns dict(left, right: uint8 [n,L]; lqual, rqual: uint8 [n,L])."""
    genome, tstart, tlen = make_transcriptome(G, seed)
    rng = np.random.default_rng(seed + 1)
    w = np.ones(len(tstart)) if uniform_expr else rng.lognormal(0.0, sigma, len(tstart))
    w = w * tlen
    w /= w.sum()
    t = rng.choice(len(tstart), n_pairs, p=w)
    flen = np.clip(np.rint(rng.normal(frag_mean, frag_sd, n_pairs)).astype(np.int64), L, None)
    flen = np.minimum(flen, tlen[t])
    fstart = tstart[t] + (rng.random(n_pairs) * (tlen[t] - flen + 1)).astype(np.int64)
    ar = np.arange(L, dtype=np.int64)
    left = genome[fstart[:, None] + ar[None, :]]
    right_fwd = genome[(fstart + flen - L)[:, None] + ar[None, :]]
    right = revcomp(right_fwd)
    out = {}
    for name, reads in (("left", left), ("right", right)):
        reads = reads.copy()
        qual = np.full(reads.shape, ord("I"), np.uint8)
        if err > 0:
            e = rng.random(reads.shape) < err
            sub = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, reads.shape)]
            same = sub == reads
            sub[same] = np.frombuffer(b"CGTA", np.uint8)[np.searchsorted(np.frombuffer(b"ACGT", np.uint8), reads[same])]
            reads[e] = sub[e]
            qual[e] = ord("#")
        if n_rate > 0:
            nn = rng.random(reads.shape) < n_rate
            reads[nn] = ord("N")
        out[name] = reads
        out[name[0] + "qual"] = qual
    return out


def flat(reads2d):
    """[n,L] uint8 -> (flat uint8, int64 offsets[n+1])"""
    n, L = reads2d.shape
    return np.ascontiguousarray(reads2d).reshape(-1), np.arange(n + 1, dtype=np.int64) * L
