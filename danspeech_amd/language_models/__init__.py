"""Plugin surface ``danspeech.language_models`` (reference danspeech/language_models/*.py).

Every factory returns the *path* of a language model file in the cache
(``~/.danspeech/lms/<file>``), exactly what the reference hands to
``Recognizer.update_decoder(lm=...)`` (reference dsl_3gram.py:7-20).  ``dsmi_decoder_set_lm``
(csrc/decoder.hip) tells the file type by its first bytes and reads KenLM binaries
(``.klm``: data structures ``probing`` -- build_binary's default -- and ``trie``, unquantised;
csrc/lm_klm.cpp.inc) as well as ARPA text; the quantised / array-compressed trie variants are
refused by name.  No ``.klm`` is obtainable offline, so the reader is checked against a
restatement of KenLM's layout (oracle/klm.py), not against KenLM's own files.
"""
from ..utils.data_utils import get_model

_RELEASE = "https://github.com/danspeech/danspeech/releases/download/v0.02-alpha/"

REGISTRY = {
    "DSL3gram": ("dsl_3gram.klm", "33ca3e2a8db3a036af6d7ad85972dbb0"),
    "DSL5gram": ("dsl_5gram.klm", "f2929d6d154b57b8be0c05347036c7e6"),
    "DSL3gramWithNames": ("dsl_names.klm", "1b47e2db841c6be5c62004ef51a40c68"),
    "DSLWiki3gram": ("dsl_wiki_3gram.klm", "f38f55a1e14ad888cee3ea1e643593dc"),
    "DSLWiki5gram": ("dsl_wiki_5gram.klm", "070287617eacbbde79df2be34ac9615f"),
    "DSLWikiLeipzig3gram": ("dsl_wiki_leipzig_3gram.klm", "8409a469be718209afdd18692a2d5609"),
    "Wiki3gram": ("wiki_3gram.klm", "12877123bbbbaa72826746cad0af6f7d"),
    "Wiki5gram": ("wiki_5gram.klm", "b329e215b2fde5ffe3e2c94204f6c189"),
    "Folketinget3gram": ("da_lm_3gram_folketinget.klm", "011771d8bef6ff531812a768f631b4a2"),
}


def _make(name):
    fname, md5 = REGISTRY[name]

    def factory(cache_dir=None):
        return get_model(model_name=fname, origin=_RELEASE + fname, file_hash=md5, cache_dir=cache_dir,
                         file_type="language_model")

    factory.__name__ = name
    factory.__doc__ = ("Path to the %s language model (%s).\n\n:param str cache_dir: custom cache directory "
                       "(default ``~/.danspeech/lms/``).\n:rtype: str" % (name, fname))
    return factory


DSL3gram = _make("DSL3gram")
DSL5gram = _make("DSL5gram")
DSL3gramWithNames = _make("DSL3gramWithNames")
DSLWiki3gram = _make("DSLWiki3gram")
DSLWiki5gram = _make("DSLWiki5gram")
DSLWikiLeipzig3gram = _make("DSLWikiLeipzig3gram")
Wiki3gram = _make("Wiki3gram")
Wiki5gram = _make("Wiki5gram")
Folketinget3gram = _make("Folketinget3gram")


def CustomLanguageModel(path):
    """Identity wrapper, as in the reference (custom_lm.py:3-14)."""
    return path
