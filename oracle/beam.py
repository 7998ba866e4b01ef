"""CTC prefix beam search with an optional word-level n-gram scorer -- TEST INFRASTRUCTURE
(see oracle/__init__.py).  PARITY UNPINNED.

What the reference runs for ``BeamCTCDecoder.decode`` (danspeech/deepspeech/decoder.py:129-144)
is the third-party ``ctcdecode.CTCBeamDecoder`` (parlance/ctcdecode, unpinned master, absent
from /root/reference; constructed at decoder.py:99-100 with log_probs_input=False).  This file
restates that package's published algorithm (ctc_beam_search_decoder.cpp / path_trie.cpp /
scorer.cpp, PaddlePaddle DeepSpeech lineage) in Python, structure for structure:

* per frame: optional vocabulary pruning (cutoff_prob / cutoff_top_n), log(p + FLT_MIN);
* with a scorer: beams sorted, ``min_cutoff = worst score + log p(blank) - max(0, beta)`` and
  the per-character early exit once ``log p(c) + score < min_cutoff`` on a full beam;
* blank extends ``p_b`` of the prefix, a repeated character extends ``p_nb`` of the prefix from
  ``p_nb`` and of prefix+c only from ``p_b``, any other character extends prefix+c from the total;
* word LM: on the space character add ``alpha * ln P(word | history) + beta``; prefixes are
  confined to the LM vocabulary by a dictionary over the trie (a new child is refused when the
  dictionary has no such arc; the dictionary state restarts after a completed word);
* keep the ``beam_size`` best by (score desc, last character asc); at the end add the LM score
  of a trailing partial word, sort, and report score = -(total - len*beta - alpha*sentence LM).

One deliberate difference: ctcdecode keeps the trie's log-probabilities in C ``float``; they are
carried in float64 here and on the GPU so that the two sides of the parity test agree to ~1e-12.
The deviation from a float implementation is its own rounding noise (~6e-5 at |score| ~ 1e3).
Ties between equal scores and equal last characters are broken by creation order.
"""
import math

from oracle.lm import Scorer  # noqa: F401  (re-exported for tests)

NEG_INF = -math.inf
FLT_MIN = 1.17549435e-38      # std::numeric_limits<float>::min(), decoder_utils.h NUM_FLT_MIN


def log_sum_exp(x, y):
    if x == NEG_INF:
        return y
    if y == NEG_INF:
        return x
    m = max(x, y)
    return math.log(math.exp(x - m) + math.exp(y - m)) + m


class PathTrie:
    _counter = 0

    def __init__(self):
        self.log_prob_b_prev = NEG_INF
        self.log_prob_nb_prev = NEG_INF
        self.log_prob_b_cur = NEG_INF
        self.log_prob_nb_cur = NEG_INF
        self.log_prob_c = NEG_INF
        self.score = NEG_INF
        self.approx_ctc = NEG_INF
        self.character = -1        # ROOT
        self.timestep = 0
        self.exists = True
        self.parent = None
        self.children = []         # list of (char, PathTrie), insertion order
        self.dict_state = 0        # node of the scorer's character trie
        self.has_dictionary = False
        self.scorer = None
        PathTrie._counter += 1
        self.uid = PathTrie._counter

    def get_path_trie(self, new_char, new_timestep, cur_log_prob_c, reset=True):
        """path_trie.cpp get_path_trie."""
        for ch, child in self.children:
            if ch == new_char:
                if child.log_prob_c < cur_log_prob_c:
                    child.log_prob_c = cur_log_prob_c
                    child.timestep = new_timestep
                if not child.exists:
                    child.exists = True
                    child.log_prob_b_prev = NEG_INF
                    child.log_prob_nb_prev = NEG_INF
                    child.log_prob_b_cur = NEG_INF
                    child.log_prob_nb_cur = NEG_INF
                return child
        new_state = 0
        if self.has_dictionary:
            sc = self.scorer
            if new_char == sc.space_id:
                # the dictionary has a space arc only out of a state where a vocabulary word ends;
                # the state behind it is final, so the spell checker restarts at the start state
                if sc.trie_word[self.dict_state] is None:
                    return None
                new_state = 0
            else:
                nxt = sc.trie_children[self.dict_state].get(new_char)
                if nxt is None:
                    return None
                new_state = nxt
        node = PathTrie()
        node.character = new_char
        node.timestep = new_timestep
        node.parent = self
        node.has_dictionary = self.has_dictionary
        node.scorer = self.scorer
        node.dict_state = new_state
        node.log_prob_c = cur_log_prob_c
        self.children.append((new_char, node))
        return node

    def get_path_vec(self):
        out, ts = [], []
        n = self
        while n.character != -1:
            out.append(n.character)
            ts.append(n.timestep)
            n = n.parent
        return out[::-1], ts[::-1]

    def iterate_to_vec(self, output):
        if self.exists:
            self.log_prob_b_prev = self.log_prob_b_cur
            self.log_prob_nb_prev = self.log_prob_nb_cur
            self.log_prob_b_cur = NEG_INF
            self.log_prob_nb_cur = NEG_INF
            self.score = log_sum_exp(self.log_prob_b_prev, self.log_prob_nb_prev)
            output.append(self)
        for _, child in self.children:
            child.iterate_to_vec(output)

    def remove(self):
        self.exists = False
        if not self.children and self.parent is not None:
            p = self.parent
            p.children = [(c, n) for (c, n) in p.children if n is not self]
            if not p.children and not p.exists:
                p.remove()


def _sort_key(p):
    # prefix_compare: score desc, then character asc; remaining ties by creation order
    return (-p.score, p.character, p.uid)


def make_ngram(scorer, prefix):
    """scorer.cpp make_ngram for a word-level LM: the last max_order words, '<s>'-padded."""
    words = []
    node = prefix
    for order in range(scorer.max_order):
        chars = []
        while node.character != -1 and node.character != scorer.space_id:
            chars.append(node.character)
            node = node.parent
        words.append("".join(scorer.labels[c] for c in reversed(chars)))
        if node.character == -1:
            words.extend(["<s>"] * (scorer.max_order - order - 1))
            break
        node = node.parent   # skip the space
    return words[::-1]


def pruned_log_probs(prob, cutoff_prob, cutoff_top_n):
    """decoder_utils.cpp get_pruned_log_probs (log_input == False)."""
    idx = list(range(len(prob)))
    cutoff_len = len(prob)
    if cutoff_prob < 1.0 or cutoff_top_n < cutoff_len:
        idx.sort(key=lambda i: (-prob[i], i))
        if cutoff_prob < 1.0:
            cum = 0.0
            cutoff_len = 0
            for i in idx:
                cum += prob[i]
                cutoff_len += 1
                if cum >= cutoff_prob or cutoff_len >= cutoff_top_n:
                    break
        else:
            cutoff_len = cutoff_top_n
        idx = idx[:cutoff_len]
    return [(i, math.log(prob[i] + FLT_MIN)) for i in idx]


def ctc_beam_search(probs, labels, beam_size, cutoff_prob=1.0, cutoff_top_n=40, blank_id=0, scorer=None):
    """probs: [T][C] probabilities of one utterance.
    Returns a list of (score, tokens, timesteps) best first, at most beam_size long."""
    space_id = labels.index(" ") if " " in labels else -2
    PathTrie._counter = 0
    root = PathTrie()
    root.score = root.log_prob_b_prev = 0.0
    if scorer is not None:
        root.has_dictionary = True
        root.scorer = scorer
    prefixes = [root]
    for t in range(len(probs)):
        prob = [float(v) for v in probs[t]]
        min_cutoff = NEG_INF
        full_beam = False
        if scorer is not None:
            num = min(len(prefixes), beam_size)
            prefixes[:num] = sorted(prefixes[:num], key=_sort_key)
            blank_lp = math.log(prob[blank_id]) if prob[blank_id] > 0 else NEG_INF
            min_cutoff = prefixes[num - 1].score + blank_lp - max(0.0, scorer.beta)
            full_beam = (num == beam_size)
        for c, log_prob_c in pruned_log_probs(prob, cutoff_prob, cutoff_top_n):
            for i in range(min(len(prefixes), beam_size)):
                prefix = prefixes[i]
                if full_beam and log_prob_c + prefix.score < min_cutoff:
                    break
                if c == blank_id:
                    prefix.log_prob_b_cur = log_sum_exp(prefix.log_prob_b_cur, log_prob_c + prefix.score)
                    continue
                if c == prefix.character:
                    prefix.log_prob_nb_cur = log_sum_exp(prefix.log_prob_nb_cur, log_prob_c + prefix.log_prob_nb_prev)
                new = prefix.get_path_trie(c, t, log_prob_c)
                if new is not None:
                    log_p = NEG_INF
                    if c == prefix.character and prefix.log_prob_b_prev > NEG_INF:
                        log_p = log_prob_c + prefix.log_prob_b_prev
                    elif c != prefix.character:
                        log_p = log_prob_c + prefix.score
                    if scorer is not None and c == space_id:
                        score = scorer.get_log_cond_prob(make_ngram(scorer, prefix)) * scorer.alpha
                        log_p += score
                        log_p += scorer.beta
                    new.log_prob_nb_cur = log_sum_exp(new.log_prob_nb_cur, log_p)
        prefixes = []
        root.iterate_to_vec(prefixes)
        if len(prefixes) >= beam_size:
            prefixes.sort(key=_sort_key)          # std::nth_element: only the top set matters
            for p in prefixes[beam_size:]:
                p.remove()
            prefixes = prefixes[:beam_size]
    if scorer is not None:
        for p in prefixes[:beam_size]:
            if p.character != -1 and p.character != space_id:
                score = scorer.get_log_cond_prob(make_ngram(scorer, p)) * scorer.alpha
                score += scorer.beta
                p.score += score
    prefixes = sorted(prefixes[:beam_size], key=_sort_key)
    out = []
    for p in prefixes:
        tokens, steps = p.get_path_vec()
        approx = p.score
        if scorer is not None:
            s = "".join(labels[c] for c in tokens)
            words = [w for w in s.split(" ") if w]
            approx = approx - len(tokens) * scorer.beta
            approx -= scorer.get_sent_log_prob(words) * scorer.alpha
        out.append((-approx, tokens, steps))
    return out


def beam_decode(probs, sizes, labels, beam_width, lm_path=None, alpha=0.0, beta=0.0, cutoff_top_n=40,
                cutoff_prob=1.0, blank_index=0, scorer=None):
    """BeamCTCDecoder.decode restated: -> (strings[B][beam], offsets[B][beam], scores[B][beam])."""
    if scorer is None and lm_path:
        scorer = Scorer(alpha, beta, lm_path, labels)
    strings, offsets, scores = [], [], []
    for b in range(len(probs)):
        n = int(sizes[b]) if sizes is not None else len(probs[b])
        res = ctc_beam_search(probs[b][:n], labels, beam_width, cutoff_prob, cutoff_top_n, blank_index, scorer)
        strings.append(["".join(labels[c] for c in r[1]) for r in res] + [""] * (beam_width - len(res)))
        offsets.append([list(r[2]) for r in res] + [[] for _ in range(beam_width - len(res))])
        scores.append([r[0] for r in res] + [0.0] * (beam_width - len(res)))
    return strings, offsets, scores
