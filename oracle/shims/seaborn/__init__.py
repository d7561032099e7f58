"""Empty stand-in: plotting is out of scope (oracle/shims/README.md)."""


def histplot(*a, **k):  # pragma: no cover
    return None
