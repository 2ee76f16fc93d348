"""One graph handle, many host threads (SURVEY §8(b) threading contract): the reference's stage-2 workers call
graph.contains / getCount / getKmers / getSuccessors on one BloomFilterDeBruijnGraph from T threads.  Queries lease their
own stream + scratch and run side by side; calls that change a filter take the handle exclusively."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom import synth
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch


def test_eight_threads_query_one_handle_while_pairs_are_inserted():
    d = synth.generate_pairs(3000, G=20000, err=0.003, n_rate=1e-3, seed=31)
    sizes = (900_007, 1_500_007, 300_007)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 5)
    gg = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, 25, False, True, rngSeed=5)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, 0); gg.addReads(s, q, off, 3)
    reads = [bytes(s[off[i]:off[i + 1]]) for i in range(400)]
    # expected answers, from the oracle, per thread
    T = 8
    work = []
    for t in range(T):
        mine = reads[t::T]
        exp = []
        for sq in mine:
            f, r, c = og.get_kmers(sq)
            exp.append((f, r, c))
        work.append((mine, exp))
    errors, done = [], threading.Event()

    def querier(t):
        try:
            mine, exp = work[t]
            for rep in range(6):
                ko, f, r, c = gg.getKmers(mine)
                for i, (ef, er, ec) in enumerate(exp):
                    a, b = ko[i], ko[i + 1]
                    assert (f[a:b] == ef).all() and (r[a:b] == er).all() and (c[a:b] == ec).all()
                base = np.where(r.view(np.int64) < f.view(np.int64), r, f)
                clean = np.concatenate([np.full(ko[i + 1] - ko[i], b"N" not in sq, bool) for i, sq in enumerate(mine)])   # getKmers zeroes the count of windows with an N
                cnt = gg.getCount(base)
                assert (gg.contains(base) == (cnt > 0)).all()
                assert (cnt[clean] == c[clean]).all()
                first = np.concatenate([np.frombuffer(sq[:len(sq) - 24], np.uint8) for sq in mine])
                f4, r4, c4 = gg.getNeighbors(f[:500], r[:500], first[:500], 0)
                for i in range(0, 500, 50):
                    of4, or4, oc4 = og.neighbors(f[i], r[i], int(first[i]), 0)
                    assert (f4[i] == of4).all() and (c4[i] == oc4).all()
                walk = gg.walkMaxCov([sq[:25] for sq in mine[:20] if b"N" not in sq[:25]], 0, 40)
                assert len(walk[0]) > 0
        except Exception as e:      # noqa: BLE001
            errors.append((t, repr(e)))

    def inserter():
        # order-free pair inserts into rpkbf: they take the handle exclusively and leave dbgbf / cbf (what the queries read) alone
        sr, offr = synth.flat(d["right"]); qr, _ = synth.flat(d["rqual"])
        batch = ReadBatch.from_ascii(sr, qr, offr, 3)
        n = 0
        while not done.is_set() and n < 40:
            gg.addPairs(batch, N.RPKBF, reverseComplement=True)
            n += 1

    th = [threading.Thread(target=querier, args=(t,)) for t in range(T)] + [threading.Thread(target=inserter)]
    for x in th: x.start()
    for x in th[:T]: x.join()
    done.set(); th[T].join()
    assert not errors, errors
    # and the state is what a single-threaded run leaves
    assert (gg.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all() and (gg.exportFilter(N.CBF) == og.cbf_bytes()).all()
