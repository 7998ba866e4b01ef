"""The beam-search KERNEL's formulation of oracle/beam.py, restated on the CPU -- TEST INFRASTRUCTURE
(see oracle/__init__.py).  PARITY UNPINNED, like oracle/beam.py, which it must equal.

`danspeech_amd/csrc/decoder.hip` does not keep ctcdecode's pointer trie.  The beam lives in on-chip arrays
and the prefix trie is implicit:

* a beam entry j carries its node, last character, dictionary state, depth and CTC terms, plus one tuple
  `(up, upch, upnode, uplpc)`: `up` = slot of its nearest ancestor that is ITSELF in the beam (-1: none),
  `upch` / `upnode` = the child of that ancestor on the path to j (the "top" of the edge; j itself when its
  parent is in the beam: a DIRECT entry) and that node's `log_prob_c`;
* "PathTrie::exists" = being a beam entry; a node is ALIVE (not removed) iff it lies on the path of some beam
  entry.  Only children of beam entries can be touched by a frame, and the alive children of entry P are
  exactly the tops `{(upch, upnode) of j : up[j] = P}`: a table cell[P][c] -> any such j answers
  `get_path_trie`.  No child counts, no cascading `remove()`: a node nobody's tuple reaches is gone, and a later
  extension to the same prefix makes a fresh node (new timestep, new log_prob_c), which is what removal means;
* when P leaves the beam, the entries below it inherit P's tuple (P has become part of their edge); when a
  dormant top D comes back (the candidate (P, c) of a cell whose entries are all indirect is selected), the
  entries of that cell hang under D, and their new top is found by walking up their parent chain in the node
  pool (the only read of the pool inside the frame loop; the depth difference says how far);
* nodes are records in a pool (parent, ch, tstep, lpc), written once at creation, tstep/lpc updated in place
  (`get_path_trie`'s "better emission frame" rule), read only by that walk and by the final path output.

This file is that algorithm, array for array and phase for phase (F1 candidates, selection, commit), in plain
Python, so that the data-structure invariants can be tested against oracle/beam.py without a GPU
(tests/test_oracle_beam_flat.py).  Reference call site: danspeech/deepspeech/decoder.py:129-144.
"""
import math

from oracle.beam import FLT_MIN, NEG_INF, log_sum_exp, pruned_log_probs
from oracle.lm import LOG10_E, OOV_SCORE, START_TOKEN

stats = {"revivals": 0, "walk_hops": 0, "frames": 0, "inherit_hops": 0}
KERNEL_THREADS = 1024      # the kernel numbers the next beam's slots wave by wave, round by round (see _slot_order)


def _slot_order(idx, nb, beam_size, n_labels, space_id, has_lm):
    """Where the kernel's thread layout puts candidate ``idx`` in its (wave, round, lane) numbering of the next beam: the
    first ceil(beam / 64) waves carry the entries (candidate j on thread j); with a scorer the next ones the (entry, space)
    pairs; the other threads the remaining pairs, dealt round robin."""
    if idx < nb:
        return (idx // 64, 0, idx % 64)
    ew = (beam_size + 63) // 64
    p = idx - nb
    if has_lm and p % n_labels == space_id:
        tid = 64 * ew + p // n_labels
        return (tid // 64, 0, tid % 64)
    rw = 2 * ew if has_lm else ew
    pt = KERNEL_THREADS - 64 * rw
    tid, k = 64 * rw + p % pt, p // pt
    return (tid // 64, k, tid % 64)


class _Entries:
    FIELDS = ("node", "ch", "ds", "depth", "bprev", "nbprev", "score", "up", "upch", "upnode", "uplpc", "ownlpc", "ctx")

    def __init__(self):
        for f in self.FIELDS:
            setattr(self, f, [])

    def n(self):
        return len(self.node)

    def push(self, **kw):
        for f in self.FIELDS:
            getattr(self, f).append(kw[f])

    def direct(self, j):
        return self.upnode[j] == self.node[j]


def ctc_beam_search(probs, labels, beam_size, cutoff_prob=1.0, cutoff_top_n=40, blank_id=0, scorer=None):
    C = len(labels)
    space_id = labels.index(" ") if " " in labels else -2
    has_lm = scorer is not None
    nctx = scorer.max_order - 1 if has_lm else 0
    # node pool: parallel lists
    n_parent, n_ch, n_tstep, n_lpc = [-1], [-1], [0], [NEG_INF]
    memo = {}          # (node id) -> ln P_lm of the word ending at that node (the kernel memoises per beam slot)

    def trie_arc(ds, c):
        """dictionary arc out of state ds: space -> the word ending here (or None), else the next state (or None)."""
        if c == space_id:
            return scorer.trie_word[ds]
        return scorer.trie_children[ds].get(c)

    E = _Entries()
    E.push(node=0, ch=-1, ds=0, depth=0, bprev=0.0, nbprev=NEG_INF, score=0.0, up=-1, upch=-1, upnode=-1, uplpc=NEG_INF,
           ownlpc=NEG_INF, ctx=[START_TOKEN] * nctx)
    cell = {}

    for t in range(len(probs)):
        stats["frames"] += 1
        prob = [float(v) for v in probs[t]]
        pr = pruned_log_probs(prob, cutoff_prob, cutoff_top_n)
        use = [False] * C
        lp = [NEG_INF] * C
        for c, l in pr:
            use[c] = True
            lp[c] = l
        nb = E.n()
        full, mincut = False, NEG_INF
        if has_lm:
            blank_lp = math.log(prob[blank_id]) if prob[blank_id] > 0 else NEG_INF
            mincut = min(E.score) + blank_lp - max(0.0, scorer.beta)
            full = nb == beam_size

        def passes(i, c):
            return use[c] and not (full and lp[c] + E.score[i] < mincut)

        def pair_logp(P, c):
            l = lp[c]
            if c == E.ch[P]:
                logp = l + E.bprev[P] if E.bprev[P] > NEG_INF else NEG_INF
            else:
                logp = l + E.score[P]
            if has_lm and c == space_id:
                w = scorer.trie_word[E.ds[P]]
                lm = OOV_SCORE
                if w is not None:
                    key = E.node[P]
                    if key not in memo:
                        memo[key] = scorer.lm.cond_log10(E.ctx[P] + [w]) / LOG10_E
                    lm = memo[key]
                logp += lm * scorer.alpha
                logp += scorer.beta
            return logp

        # ---- F1: entries (pull form) and pairs -> candidates (key, char, index, kind, payload)
        cands = []
        bcur, nbcur = [NEG_INF] * nb, [NEG_INF] * nb
        for j in range(nb):
            b = r = x = NEG_INF
            if passes(j, blank_id):
                b = lp[blank_id] + E.score[j]
            cj = E.ch[j]
            if cj >= 0 and cj != blank_id and passes(j, cj):
                r = lp[cj] + E.nbprev[j]
            P = E.up[j]
            if P >= 0 and passes(P, E.upch[j]):
                c = E.upch[j]
                if E.uplpc[j] < lp[c]:                 # get_path_trie: a better emission frame for the edge's top
                    E.uplpc[j] = lp[c]
                    n_lpc[E.upnode[j]] = lp[c]
                    n_tstep[E.upnode[j]] = t
                if E.direct(j):
                    x = pair_logp(P, c)
            # the kernel's form of nb_cur = lse(r, x), score = lse(b, nb_cur): both exps and both logs independent of each other
            m2 = max(r, x)
            e1 = math.exp(min(r, x) - m2) if (r > NEG_INF and x > NEG_INF) else 0.0
            e2 = math.exp(-abs(b - m2)) if (b > NEG_INF and m2 > NEG_INF) else 0.0
            sum2 = 1.0 + e1
            xx = 1.0 + sum2 * e2 if b >= m2 else e2 + sum2
            nbcur[j] = m2 + math.log(sum2)
            bcur[j] = b
            s = max(b, m2) + math.log(xx)
            cands.append((s, E.ch[j], j, "stay", j))
        for i in range(nb):
            for c in range(C):
                if c == blank_id or not passes(i, c):
                    continue
                rep = cell.get((i, c), -1)
                arc = None
                if rep >= 0:
                    if E.direct(rep):
                        continue                        # the child is a beam entry: it pulled this contribution itself
                    kind = "revive"
                else:
                    kind = "fresh"
                    if has_lm:
                        arc = trie_arc(E.ds[i], c)
                        if arc is None:
                            continue                    # the dictionary has no such arc
                cands.append((pair_logp(i, c), c, nb + i * C + c, kind, (i, c, rep)))

        # ---- selection: top beam_size by (score desc, char asc, candidate index asc)
        order = sorted(range(len(cands)), key=lambda q: (-cands[q][0], cands[q][1], cands[q][2]))
        keep = set(order[:beam_size])
        # new slots in the kernel's order (which only matters for how later exact ties fall)
        newslot = [-1] * nb
        pos = 0
        sel = []
        for q in sorted(keep, key=lambda q: _slot_order(cands[q][2], nb, beam_size, C, space_id, has_lm)):
            sel.append((q, pos))
            if cands[q][3] == "stay":
                newslot[cands[q][4]] = pos
            pos += 1
        revived = {}          # (P, c) -> new slot of the revived top
        for q, p in sel:
            if cands[q][3] == "revive":
                i, c, rep = cands[q][4]
                revived[(i, c)] = p

        # ---- commit
        N = _Entries()

        def resolve(tup, node_j, depth_j, ownlpc_j):
            """(up, upch, upnode, uplpc) in OLD slot numbers -> the tuple in new slot numbers."""
            up, upch, upnode, uplpc = tup
            while up >= 0:
                if (up, upch) in revived and upnode != node_j:
                    # the dormant top of this edge is a beam entry again: hang under it
                    slot = revived[(up, upch)]
                    top_depth = E.depth[up] + 1
                    n = node_j
                    hops = depth_j - top_depth - 1
                    for _ in range(hops):
                        n = n_parent[n]
                    stats["walk_hops"] += max(hops, 0)
                    assert n_parent[n] == upnode, "walk must end right below the revived top"
                    lpc = ownlpc_j if n == node_j else n_lpc[n]
                    return slot, n_ch[n], n, lpc
                if newslot[up] >= 0:
                    return newslot[up], upch, upnode, uplpc
                # the ancestor left the beam: it is part of this edge now
                stats["inherit_hops"] += 1
                up, upch, upnode, uplpc = E.up[up], E.upch[up], E.upnode[up], E.uplpc[up]
            return -1, -1, -1, NEG_INF

        for q, p in sel:
            s, ch, idx, kind, pay = cands[q]
            if kind == "stay":
                j = pay
                own = E.uplpc[j] if E.direct(j) else E.ownlpc[j]
                up, upch, upnode, uplpc = resolve((E.up[j], E.upch[j], E.upnode[j], E.uplpc[j]), E.node[j], E.depth[j], own)
                N.push(node=E.node[j], ch=E.ch[j], ds=E.ds[j], depth=E.depth[j], bprev=bcur[j], nbprev=nbcur[j], score=s,
                       up=up, upch=upch, upnode=upnode, uplpc=uplpc, ownlpc=own, ctx=E.ctx[j])
                continue
            i, c, rep = pay
            if kind == "fresh":
                nid = len(n_parent)
                n_parent.append(E.node[i]); n_ch.append(c); n_tstep.append(t); n_lpc.append(lp[c])
                lpc = lp[c]
            else:
                stats["revivals"] += 1
                nid = E.upnode[rep]
                lpc = E.uplpc[rep]
            nds, ctx = 0, E.ctx[i]
            if has_lm:
                arc = trie_arc(E.ds[i], c)
                if c == space_id:
                    nds = 0
                    ctx = (E.ctx[i] + [arc])[1:] if nctx > 0 else []
                else:
                    nds = arc
            # raw tuple (parent i, c, itself), then resolved like everybody's (the parent may have left the beam)
            up, upch, upnode, uplpc = resolve_new(i, c, nid, lpc, E, newslot, revived, resolve)
            N.push(node=nid, ch=c, ds=nds, depth=E.depth[i] + 1, bprev=NEG_INF, nbprev=s, score=s, up=up, upch=upch,
                   upnode=upnode, uplpc=uplpc, ownlpc=lpc, ctx=ctx)
        E = N
        cell = {}
        for j in range(E.n()):
            if E.up[j] >= 0:
                key = (E.up[j], E.upch[j])
                if key in cell:
                    assert not E.direct(j) and not E.direct(cell[key]), "a direct child excludes indirect ones"
                    assert E.upnode[cell[key]] == E.upnode[j]
                cell[key] = j

    # ---- final: trailing partial word, order, paths
    final = []
    for j in range(E.n()):
        s = E.score[j]
        if has_lm and E.ch[j] != -1 and E.ch[j] != space_id:
            w = scorer.trie_word[E.ds[j]]
            lm = OOV_SCORE if w is None else scorer.lm.cond_log10(E.ctx[j] + [w]) / LOG10_E
            sc = lm * scorer.alpha
            sc += scorer.beta
            s += sc
        final.append((s, E.ch[j], j))
    final.sort(key=lambda f: (-f[0], f[1], f[2]))
    out = []
    for s, _, j in final:
        tokens, steps = [], []
        n = E.node[j]
        while n > 0:
            tokens.append(n_ch[n]); steps.append(n_tstep[n])
            n = n_parent[n]
        tokens.reverse(); steps.reverse()
        approx = s
        if has_lm:
            txt = "".join(labels[c] for c in tokens)
            words = [w for w in txt.split(" ") if w]
            approx = approx - len(tokens) * scorer.beta
            approx -= scorer.get_sent_log_prob(words) * scorer.alpha
        out.append((-approx, tokens, steps))
    return out


def resolve_new(i, c, nid, lpc, E, newslot, revived, resolve):
    """A new entry's raw tuple is (its parent's old slot, c, itself); a revived top must not mistake its own cell's
    'revived' mark for an edge above it."""
    if newslot[i] >= 0:
        return newslot[i], c, nid, lpc
    # parent left the beam in this very frame: inherit its tuple, then resolve that
    return resolve((E.up[i], E.upch[i], E.upnode[i], E.uplpc[i]), nid, E.depth[i] + 1, lpc)
