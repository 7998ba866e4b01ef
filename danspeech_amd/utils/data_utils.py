"""Artefact cache of the plugin surface: where ``pretrained_models.*`` / ``language_models.*`` look for
their files and how a file is accepted (reference danspeech/utils/data_utils.py: same public names
``get_model`` / ``validate_file`` / ``subdir_mapper``, same layout ``~/.danspeech/{models,lms}/<file>``,
same md5 acceptance test).

Fetching the file is network I/O outside the hot path.  It is attempted only when the optional ``wget``
module is importable; unlike the reference, a file that fails its md5 is removed BEFORE the download (so the
downloader cannot park the new file beside the stale one under another name) and the downloaded file must
pass the md5 too, otherwise it is deleted and the call raises -- a model package is unpickled by the caller
and must not be accepted unverified.
"""
import hashlib
import os

subdir_mapper = {"acoustic_model": "models", "language_model": "lms"}


def _md5_of(path, chunk_size=65535):
    digest = hashlib.md5()
    with open(path, "rb") as fh:
        while True:
            block = fh.read(chunk_size)
            if not block:
                return digest.hexdigest()
            digest.update(block)


def validate_file(fpath, file_hash, chunk_size=65535):
    """True when the file's md5 equals ``file_hash``."""
    return _md5_of(fpath, chunk_size) == str(file_hash)


def _cache_path(name, file_type, cache_dir):
    root = cache_dir or os.path.join(os.path.expanduser("~"), ".danspeech", subdir_mapper[file_type])
    os.makedirs(root, exist_ok=True)
    return root, os.path.join(root, name)


def _fetch(origin, path, name, root):
    try:
        import wget
    except ImportError:
        raise RuntimeError("%s is not in the cache (%s) and cannot be downloaded here (no `wget`/network). "
                           "Place the file there, or use CustomModel/CustomLanguageModel with a local path."
                           % (name, root))
    print("Downloading data from", origin)
    try:
        wget.download(url=origin, out=path)
    except BaseException:
        if os.path.exists(path):
            os.remove(path)
        raise


def get_model(model_name, origin, file_type="acoustic_model", file_hash=None, cache_dir=None):
    """Path of ``model_name`` in the cache, fetched from ``origin`` when absent or not matching ``file_hash``."""
    root, path = _cache_path(model_name, file_type, cache_dir)
    present = os.path.exists(path)
    if present and file_hash and not validate_file(path, file_hash):
        print("The cached file %s does not match its md5 %s: it is removed and fetched again" % (path, file_hash))
        os.remove(path)
        present = False
    if not present:
        _fetch(origin, path, model_name, root)
        if file_hash and not validate_file(path, file_hash):
            os.remove(path)
            raise RuntimeError("%s downloaded from %s does not match its md5 %s" % (model_name, origin, file_hash))
    return path
