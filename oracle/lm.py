"""ARPA n-gram language model with back-off scoring -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Stands in for KenLM as used through ctcdecode's ``Scorer`` (third-party, not under
/root/reference, unpinned: reference docs_source/installation.rst:23-30; call site
danspeech/deepspeech/decoder.py:95-100).  PARITY UNPINNED.  Restated from KenLM's published
behaviour: probabilities and back-offs are stored as float32 log10 values (as KenLM's binary
format does); p(w | h) is the standard back-off recursion -- longest matching n-gram, otherwise
back-off weight of the context (0 when the context n-gram is absent) plus the shorter
history -- which is what KenLM's ``BaseScore`` returns from any state.
"""
import numpy as np

OOV_SCORE = -1000.0          # ctcdecode scorer.h: const double OOV_SCORE = -1000.0
START_TOKEN = "<s>"
END_TOKEN = "</s>"
UNK_TOKEN = "<unk>"
LOG10_E = float(np.float32(0.4342944819))   # ctcdecode decoder_utils.h: const float NUM_FLT_LOGE


class ArpaLM:
    def __init__(self, path):
        self.ngrams = {}      # tuple(words) -> (log10 prob, log10 backoff) as python floats of float32 values
        self.order = 0
        self.vocab = []
        self._parse(path)

    def _parse(self, path):
        section = 0
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                line = line.rstrip("\n")
                if not line.strip():
                    continue
                if line.startswith("\\data\\"):
                    continue
                if line.startswith("ngram "):
                    n = int(line.split()[1].split("=")[0])
                    self.order = max(self.order, n)
                    continue
                if line.startswith("\\end\\"):
                    break
                if line.startswith("\\") and line.endswith("-grams:"):
                    section = int(line[1:line.index("-")])
                    continue
                parts = line.split("\t") if "\t" in line else line.split()
                if "\t" in line:
                    lp = float(np.float32(parts[0]))
                    words = tuple(parts[1].split(" "))
                    bo = float(np.float32(parts[2])) if len(parts) > 2 else 0.0
                else:
                    lp = float(np.float32(parts[0]))
                    words = tuple(parts[1:1 + section])
                    bo = float(np.float32(parts[1 + section])) if len(parts) > 1 + section else 0.0
                self.ngrams[words] = (lp, bo)
                if section == 1:
                    self.vocab.append(words[0])
        self.vocab_set = set(self.vocab)

    def cond_log10(self, words):
        """log10 p(words[-1] | words[:-1]) by back-off; words all in vocabulary."""
        hist, w = tuple(words[:-1]), words[-1]
        acc = np.float32(0.0)                    # KenLM sums its float32 back-offs in float
        while True:
            g = hist + (w,)
            if g in self.ngrams:
                return float(np.float32(acc + np.float32(self.ngrams[g][0])))
            if not hist:
                # unigram must exist for in-vocabulary words; unknown words score as <unk>
                return float(np.float32(acc + np.float32(self.ngrams[(UNK_TOKEN,)][0])))
            if hist in self.ngrams:
                acc = np.float32(acc + np.float32(self.ngrams[hist][1]))
            hist = hist[1:]


class Scorer:
    """ctcdecode ``Scorer`` for a word-level LM (is_character_based() == False)."""

    def __init__(self, alpha, beta, lm_path, labels):
        self.alpha = float(alpha)
        self.beta = float(beta)
        self.lm = ArpaLM(lm_path)
        self.max_order = self.lm.order
        self.labels = labels
        self.space_id = labels.index(" ") if " " in labels else -2
        # dictionary: every LM word that can be spelled with the labels (scorer.cpp fill_dictionary
        # skips the words containing characters outside the vocabulary), as a character trie
        lab = {c: i for i, c in enumerate(labels)}
        self.trie_children = [{}]     # node -> {label id: node}
        self.trie_word = [None]       # node -> word string if a vocabulary word ends here
        for wd in self.lm.vocab:
            if wd in (START_TOKEN, END_TOKEN, UNK_TOKEN) or not wd:
                continue
            if any(ch not in lab or ch == " " for ch in wd):
                continue
            node = 0
            for ch in wd:
                nxt = self.trie_children[node].get(lab[ch])
                if nxt is None:
                    nxt = len(self.trie_children)
                    self.trie_children.append({})
                    self.trie_word.append(None)
                    self.trie_children[node][lab[ch]] = nxt
                node = nxt
            self.trie_word[node] = wd

    def get_log_cond_prob(self, words):
        """scorer.cpp get_log_cond_prob: natural-log p(last | previous) ; OOV anywhere -> OOV_SCORE."""
        for w in words:
            if w not in self.lm.vocab_set or w == UNK_TOKEN:
                return OOV_SCORE
        return self.lm.cond_log10(list(words)) / LOG10_E

    def get_sent_log_prob(self, words):
        """scorer.cpp get_sent_log_prob / get_log_prob."""
        if len(words) == 0:
            sentence = [START_TOKEN] * self.max_order
        else:
            sentence = [START_TOKEN] * (self.max_order - 1) + list(words)
        sentence.append(END_TOKEN)
        score = 0.0
        for i in range(len(sentence) - self.max_order + 1):
            score += self.get_log_cond_prob(sentence[i:i + self.max_order])
        return score
