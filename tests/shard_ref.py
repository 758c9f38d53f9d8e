"""Sequential, shard-aware restatement of stage C for the distributed tests (test infrastructure).

Same rules as oracle/mg_oracle.c (mgo_profile_assign, which is pinned to the reference's golden vectors),
but with an explicit incoming "first line dropped" state, an optional lookahead record that closes the
shard's last read, and global read indices — i.e. the contract of mg_profile_begin_dev / _commit_dev.
tests/test_distributed_gloo.py checks it against the C oracle on unsharded streams before relying on it.
"""
import numpy as np

NEW = 0x80000000


def _flags(fl):
    p1 = bool(fl & 1) and bool(fl & 64)
    p2 = bool(fl & 1) and bool(fl & 128)
    return p1, p2, bool(fl & 2048)


def _process(hits, recs, ref2tax, pct_id, npair):
    """-> (kind, taxon, hitlen, taxa_list); kind 0 Ambiguous, 1 unique, 2 multimapped."""
    p1 = p2 = 0
    kept, hitlen = [], 0
    for i in hits:
        fl = int(recs["flag_len"][i]) & 0xFFF
        a, b, chim = _flags(fl)
        p1 += 1 if (a or not (a or b)) else 0
        p2 += 1 if b else 0
        if float(recs["matched"][i]) / float(recs["total"][i]) < pct_id or chim:
            if a:
                p1 -= 1
            elif b:
                p2 -= 1
        else:
            kept.append(int(ref2tax[int(recs["ref_new"][i]) & 0x7FFFFFFF]))
        hitlen += int(recs["flag_len"][i]) >> 12
    if not kept:
        return 0, 0, hitlen, []
    if npair[0] or npair[1]:
        if p1 + p2 == 1:
            return 1, kept[0], hitlen, []
        if p1 == 0 or p2 == 0:
            return 0, 0, hitlen, []
        first, second = kept[:max(p1, 0)], kept[max(p1, 0):]
        both = {t for t in first if t in second}
        if len(both) == 0:
            return 0, 0, hitlen, []
        if len(both) == 1:
            return 1, kept[0], hitlen, []
        return 2, 0, hitlen, [t for t in kept if t in both]
    if p1 > 1:
        return 2, 0, hitlen, list(kept)
    return 1, kept[0], hitlen, []


def run_shard(recs, nrecs, has_look, ref2tax, ntax, pct_id, incoming, first_shard, group_base):
    count = np.zeros(ntax, dtype=np.uint64)
    bases = np.zeros(ntax, dtype=np.uint64)
    first = np.full(ntax, 2**64 - 1, dtype=np.uint64)
    groups = ambig = 0
    outgoing = incoming if nrecs == 0 else 0
    mm = []
    hits = []
    ntotal = nrecs + (1 if has_look else 0)
    for i in range(ntotal):
        if int(recs["ref_new"][i]) & NEW:
            fl = int(recs["flag_len"][i]) & 0xFFF
            npair = _flags(fl)[:2]
            if i == 0:
                groups += 1
                if first_shard:
                    ambig += 1  # the phantom boundary: an empty read is Ambiguous
                if incoming:
                    continue    # this line is dropped
            else:
                kind, tax, hitlen, taxa = _process(hits, recs, ref2tax, pct_id, npair)
                ridx = group_base + groups - 1
                hits = []
                if kind == 1:
                    count[tax] += np.uint64(1)
                    bases[tax] += np.uint64(hitlen)
                    first[tax] = min(first[tax], np.uint64(ridx))
                elif kind == 2:
                    mm.append((ridx, taxa, hitlen))
                else:
                    ambig += 1
                if i == nrecs:  # the lookahead only closes the last read
                    outgoing = 1 if kind == 0 else 0
                    break
                groups += 1
                if kind == 0:
                    continue
        if i < nrecs:
            hits.append(i)
    return dict(count=count, bases=bases, first_seen=first, groups=groups, ambig=ambig, outgoing=outgoing, mm=mm)
