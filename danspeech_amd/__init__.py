"""danspeech_amd: the DanSpeech ``recognize()`` hot path, MI355X-native.

Drop-in for ``danspeech`` on that path: ``from danspeech_amd import Recognizer`` (reference
danspeech/__init__.py:5-6).  Imports are lazy so that host-only pieces (synthetic data, the
sharding plan, the ctypes prototypes) work without torch or a GPU.
"""
import os
import shutil
import warnings

# (Importing the package changes nothing in the process: the one runtime setting the batch pipeline wants, GPU_MAX_HW_QUEUES, is
# asked for when an engine is made -- _native.want_hw_queues.)


class NoDefaultCacheDirForDanspeech(Warning):
    pass


def clean_cache():
    """reference danspeech/__init__.py:13-22."""
    cache_dir = os.path.join(os.path.expanduser('~'), '.danspeech')
    if os.path.isdir(cache_dir):
        shutil.rmtree(cache_dir)
    else:
        warnings.warn("The default cache dir for danspeech (~.danspeech/ did not exist. If you are"
                      "using custom cache dir, then delete it manually.", NoDefaultCacheDirForDanspeech)


def __getattr__(name):
    if name in ("Recognizer", "DanSpeechRecognizer"):
        import importlib
        # as in the reference, the class shadows its module on the package
        for n in ("DanSpeechRecognizer", "Recognizer"):
            globals()[n] = getattr(importlib.import_module("." + n, __name__), n)
        return globals()[name]
    raise AttributeError(name)
