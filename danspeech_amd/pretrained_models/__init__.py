"""Plugin surface ``danspeech.pretrained_models`` (reference danspeech/pretrained_models/*.py).

Zero-argument factories with the reference's names; each resolves its released ``.pth`` package
in the cache (``~/.danspeech/models/<file>``, md5-checked: reference
danspeech/utils/data_utils.py:59-77) and returns a ``DeepSpeech`` whose arithmetic runs in
libdsmi.so.  One registry table instead of one module per artefact.
"""
from ..deepspeech.model import DeepSpeech
from ..utils.data_utils import get_model

_RELEASE = "https://github.com/danspeech/danspeech/releases/download/v0.01-alpha/"

# factory name -> (file, md5, documented shape)   [reference danspeech/pretrained_models/<x>.py]
REGISTRY = {
    "DanSpeechPrimary": ("DanSpeechPrimary.pth", "5bd08282d442e990c37481d5c61cf93c", "3 conv, 9 RNN x 1200"),
    "TestModel": ("TestModel.pth", "c21438a33f847a9c8d4e08779e98bf31", "2 conv, 5 RNN x 400"),
    "Baseline": ("Baseline.pth", "e2c0c16d518fc57cd61c86cbb0170660", "2 conv, 5 RNN x 800"),
    "TransferLearned": ("TransferLearned.pth", "d19b9d7dc976bffbc9225e0f80ecacbf", "2 conv, 5 RNN x 800"),
    "Folketinget": ("Folketinget.pth", "9523d5744ad4ff5ffc8519393350cc91", "3 conv, 9 RNN x 1200"),
    "EnglishLibrispeech": ("Librispeech.pth", "56630094905e7308f42ae0f82421440b", "2 conv, 5 RNN x 800"),
    "CPUStreamingRNN": ("CPUStreamingRNN.pth", "ba514ec96b511c0797dc643190a80269", "2 conv, 5 x 800 unidirectional, context 20"),
    "GPUStreamingRNN": ("GPUStreamingRNN.pth", "8194f47f5c63c14c3587d42aa37d622d", "2 conv, 5 x 2000 unidirectional, context 20"),
}


def _make(name):
    fname, md5, shape = REGISTRY[name]

    def factory(cache_dir=None):
        path = get_model(model_name=fname, origin=_RELEASE + fname, file_hash=md5, cache_dir=cache_dir)
        return DeepSpeech.load_model(path)

    factory.__name__ = name
    factory.__doc__ = ("Pretrained DanSpeech model %s (%s).\n\n:param str cache_dir: custom cache directory "
                       "(default ``~/.danspeech/models/``).\n:rtype: DeepSpeech" % (name, shape))
    return factory


DanSpeechPrimary = _make("DanSpeechPrimary")
TestModel = _make("TestModel")
Baseline = _make("Baseline")
TransferLearned = _make("TransferLearned")
Folketinget = _make("Folketinget")
EnglishLibrispeech = _make("EnglishLibrispeech")
CPUStreamingRNN = _make("CPUStreamingRNN")
GPUStreamingRNN = _make("GPUStreamingRNN")


def CustomModel(model_path):
    """Custom trained model from a local ``.pth`` package (reference custom_model.py:4-14)."""
    return DeepSpeech.load_model(model_path)


def get_model_from_string(model_name):
    """reference pretrained_models/__init__.py:12-30, including its quirk that
    'GPUStreamingRNN' resolves to CPUStreamingRNN (line 21-22); unknown names give None."""
    if model_name == "GPUStreamingRNN":
        return CPUStreamingRNN()
    if model_name in REGISTRY:
        return globals()[model_name]()
    return None
