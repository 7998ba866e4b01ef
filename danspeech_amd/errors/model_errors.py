"""Exception types of reference danspeech/errors/model_errors.py (same names)."""


class ConvError(Exception):
    pass


class ModelDoesNotExistError(Exception):
    pass


class FreezingMoreLayersThanExist(Exception):
    pass
