import os


def is_xformers_available():
    """True: oracle/stubs/xformers stands in.  DSU_STUB_NO_XFORMERS=1 makes the reference keep its
    default (non-xformers) processors — same arithmetic, used as a cross-check."""
    return os.environ.get("DSU_STUB_NO_XFORMERS", "0") != "1"
