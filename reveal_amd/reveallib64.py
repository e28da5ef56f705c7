"""REcursiVe Exact matching ALigner (64bit suffix array) -- drop-in for the
reference's `reveallib64` extension module (reveallib/interface.c:894-913)."""
from ._index import make_index_type


class error(Exception):                  # PyErr_NewException("Reveal.error"), interface.c:910-912
    pass


error.__name__ = "error"
error.__qualname__ = "Reveal.error"
index = make_index_type(True, error)
