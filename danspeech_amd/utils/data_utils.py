"""Cache layout and md5 validation of reference danspeech/utils/data_utils.py:7-88.

Same signatures and cache layout (``~/.danspeech/{models,lms}/<file>``).  The download step
is network I/O, not part of the hot path: when the file is absent (or fails its md5) and
``wget`` is not importable this raises instead of downloading.
"""
import hashlib
import os


def _hash_file(fpath, chunk_size=65535):
    hasher = hashlib.md5()
    with open(fpath, 'rb') as fpath_file:
        for chunk in iter(lambda: fpath_file.read(chunk_size), b''):
            hasher.update(chunk)
    return hasher.hexdigest()


def validate_file(fpath, file_hash, chunk_size=65535):
    return str(_hash_file(fpath, chunk_size)) == str(file_hash)


subdir_mapper = {"acoustic_model": "models",
                 "language_model": "lms"}


def get_model(model_name, origin, file_type="acoustic_model", file_hash=None, cache_dir=None):
    if cache_dir is None:
        cache_dir = os.path.join(os.path.expanduser('~'), '.danspeech', subdir_mapper[file_type])
    os.makedirs(cache_dir, exist_ok=True)
    download = False
    fpath = os.path.join(cache_dir, model_name)
    if os.path.exists(fpath) and file_hash:
        if not validate_file(fpath, file_hash):
            print('A local file was found, but it seems to be incomplete or outdated because the md5 '
                  'file hash does not match the original value of ' + file_hash + ' hence the model will be '
                  'redownloaded and the incomplete or outdated model will be deleted')
            download = True
    elif not os.path.exists(fpath):
        download = True
    if download:
        print('Downloading data from', origin)
        try:
            import wget
        except ImportError:
            raise RuntimeError("%s is not in the cache (%s) and cannot be downloaded here (no `wget`/network). "
                               "Place the file there, or use CustomModel/CustomLanguageModel with a local path."
                               % (model_name, cache_dir))
        try:
            wget.download(url=origin, out=fpath)
        except (Exception, KeyboardInterrupt) as e:
            if os.path.exists(fpath):
                os.remove(fpath)
            raise e
    return fpath
