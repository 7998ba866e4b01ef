"""Exception types of reference danspeech/errors/recognizer_errors.py (same names)."""


class WaitTimeoutError(Exception):
    pass


class RequestError(Exception):
    pass


class UnknownValueError(Exception):
    pass


class ModelNotInitialized(Exception):
    pass


class WrongUsageOfListen(Exception):
    pass


class NoDataInBuffer(Exception):
    pass


class ArgumentMissingForOption(Exception):
    pass
