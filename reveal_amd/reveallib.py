"""REcursiVe Exact matching ALigner -- drop-in for the reference's `reveallib`
extension module (reveallib/interface.c:917-937): exports `index` and `error`."""
from ._index import make_index_type


class error(Exception):                  # PyErr_NewException("Reveal.error"), interface.c:933-936
    pass


error.__name__ = "error"
error.__qualname__ = "Reveal.error"
index = make_index_type(False, error)
