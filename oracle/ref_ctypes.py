"""ctypes harness over oracle/_ref/libreveal_ref[64].so -- TEST INFRASTRUCTURE ONLY.

The library is the reference's own C (reveallib/reveal.c, reveallib/interface.c,
divsufsort/*.c) compiled unmodified by oracle/Makefile.  This module builds
`RevealIndex` structs (reveallib/reveal.h:17-40) around numpy arrays and calls
the reference's plain-C entry points:

    divsufsort / divsufsort64      divsufsort/divsufsort.c:326
    compute_lcp                    reveallib/interface.c:97
    build_SO                       reveallib/interface.c:116
    getmums / getmums_rem          reveallib/reveal.c:55 / :119
    getmultimums / getmultimems    reveallib/reveal.c:436 / :292
    split                          reveallib/reveal.c:582
    bubble_sort                    reveallib/reveal.c:666
    extract                        reveallib/reveal.c:1386

Nothing here is importable by the product (reveal_amd/); tests, golden-vector
generation and pinning of the CPU restatement only.  Opened RTLD_LAZY because
three Python-2-only names in never-called paths stay unresolved.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def available(sa64=False):
    return os.path.exists(os.path.join(_HERE, "_ref", "libreveal_ref64.so" if sa64 else "libreveal_ref.so"))


class Ref:
    def __init__(self, sa64=False):
        self.sa64 = sa64
        self.sa_t = np.int64 if sa64 else np.int32
        self.lcp_t = np.uint32 if sa64 else np.int32
        self.c_sa = ctypes.c_int64 if sa64 else ctypes.c_int32
        self.c_lcp = ctypes.c_uint32 if sa64 else ctypes.c_int32
        path = os.path.join(_HERE, "_ref", "libreveal_ref64.so" if sa64 else "libreveal_ref.so")
        libc = ctypes.CDLL(None)
        libc.dlopen.restype = ctypes.c_void_p
        libc.dlopen.argtypes = [ctypes.c_char_p, ctypes.c_int]
        h = libc.dlopen(path.encode(), os.RTLD_LAZY)
        if not h:
            raise OSError("cannot dlopen " + path)
        self.lib = ctypes.PyDLL(path, handle=h)
        c_sa, c_lcp = self.c_sa, self.c_lcp

        class RevealIndex(ctypes.Structure):  # reveallib/reveal.h:17-40
            _fields_ = [("ob_refcnt", ctypes.c_ssize_t), ("ob_type", ctypes.c_void_p),
                        ("T", ctypes.c_void_p), ("SA", ctypes.c_void_p), ("SAi", ctypes.c_void_p),
                        ("LCP", ctypes.c_void_p), ("SO", ctypes.c_void_p),
                        ("n", c_sa), ("nT", c_sa), ("nsep", ctypes.c_void_p),
                        ("depth", ctypes.c_int), ("nsamples", ctypes.c_int),
                        ("safile", ctypes.c_char_p), ("lcpfile", ctypes.c_char_p),
                        ("rc", ctypes.c_int), ("cache", ctypes.c_int),
                        ("main", ctypes.c_void_p), ("samples", ctypes.c_void_p), ("nodes", ctypes.c_void_p),
                        ("left_node", ctypes.c_void_p), ("right_node", ctypes.c_void_p),
                        ("skipmums", ctypes.c_void_p)]
        self.RevealIndex = RevealIndex
        L = self.lib
        self._dss = getattr(L, "divsufsort64" if sa64 else "divsufsort")
        self._dss.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_sa]
        self._dss.restype = ctypes.c_int
        L.compute_lcp.argtypes = [ctypes.c_void_p] * 4 + [c_sa]
        L.compute_lcp.restype = ctypes.c_int
        L.build_SO.argtypes = [ctypes.POINTER(RevealIndex)]
        for fn in (L.getmums, L.getmums_rem, L.getmultimums, L.getmultimems):
            fn.restype = ctypes.py_object
            fn.argtypes = [ctypes.POINTER(RevealIndex), ctypes.py_object, ctypes.py_object]
        L.split.argtypes = [ctypes.POINTER(RevealIndex), ctypes.c_void_p] + [ctypes.POINTER(RevealIndex)] * 3
        L.split.restype = None
        L.bubble_sort.argtypes = [ctypes.POINTER(RevealIndex), ctypes.py_object]
        L.bubble_sort.restype = None
        L.extract.argtypes = [ctypes.POINTER(RevealIndex), ctypes.py_object, ctypes.py_object]
        L.extract.restype = ctypes.py_object

    def divsufsort_addr(self):
        return ctypes.cast(self._dss, ctypes.c_void_p).value

    # -- construct pieces ---------------------------------------------------
    def divsufsort(self, T):
        """T: bytes / uint8 array (without trailing NUL) -> SA"""
        t = np.frombuffer(bytes(T), dtype=np.uint8) if not isinstance(T, np.ndarray) else T
        t = np.ascontiguousarray(t)
        SA = np.empty(len(t), dtype=self.sa_t)
        r = self._dss(t.ctypes.data, SA.ctypes.data, len(t))
        if r != 0:
            raise RuntimeError("divsufsort failed")
        return SA

    @staticmethod
    def textbuf(T):
        """mutable NUL-terminated copy of the text like interface.c:71-85"""
        t = np.zeros(len(T) + 1, dtype=np.uint8)
        t[:len(T)] = np.frombuffer(bytes(T), dtype=np.uint8) if not isinstance(T, np.ndarray) else T[:len(T)]
        return t

    def inverse(self, SA):
        SAi = np.empty_like(SA)
        SAi[SA] = np.arange(len(SA), dtype=SA.dtype)
        return SAi

    def compute_lcp(self, tbuf, SA, SAi):
        LCP = np.zeros(len(SA), dtype=self.lcp_t)
        self.lib.compute_lcp(tbuf.ctypes.data, SA.ctypes.data, SAi.ctypes.data, LCP.ctypes.data, len(SA))
        return LCP

    def view(self, tbuf, SA, LCP, nsep, nsamples_main, SAi=None, SO=None, nT=None, rc=0, main=None):
        """RevealIndex struct over the given arrays (kept alive on the struct)."""
        ri = self.RevealIndex()
        ri.ob_refcnt = 1 << 30
        ri.T = tbuf.ctypes.data
        ri.SA = SA.ctypes.data if SA is not None else None
        ri.LCP = LCP.ctypes.data if LCP is not None else None
        ri.SAi = SAi.ctypes.data if SAi is not None else None
        ri.SO = SO.ctypes.data if SO is not None else None
        ri.n = len(SA) if SA is not None else 0
        ri.nT = nT if nT is not None else ri.n
        nsep = np.ascontiguousarray(np.asarray(nsep, dtype=self.sa_t))
        ri.nsep = nsep.ctypes.data
        ri.nsamples = nsamples_main
        ri.rc = rc
        ri._keep = (tbuf, SA, LCP, SAi, SO, nsep, main)
        ri.main = ctypes.addressof(main if main is not None else ri)
        return ri

    def build_so(self, nsep, nsamples, n):
        SO = np.zeros(n, dtype=np.uint16)
        ri = self.RevealIndex()
        nsep = np.ascontiguousarray(np.asarray(nsep, dtype=self.sa_t))
        ri.nsep = nsep.ctypes.data
        ri.nsamples = nsamples
        ri.n = n
        ri.SO = SO.ctypes.data
        self.lib.build_SO(ctypes.byref(ri))
        return SO

    # -- scans ----------------------------------------------------------------
    def getmums(self, ri, minl):
        return self.lib.getmums(ctypes.byref(ri), (int(minl),), None)

    def getmums_rem(self, ri, minl):
        return self.lib.getmums_rem(ctypes.byref(ri), (int(minl),), None)

    def getmultimums(self, ri, minlength=0, minn=2):
        return self.lib.getmultimums(ctypes.byref(ri), (), {"minlength": int(minlength), "minn": int(minn)})

    def getmultimems(self, ri, minlength=0, minn=2):
        return self.lib.getmultimems(ctypes.byref(ri), (), {"minlength": int(minlength), "minn": int(minn)})

    # -- split / bubble -------------------------------------------------------
    def split(self, ri, D, nl, nt, np_):
        """reference split(); returns [(SA,LCP)|None]*3 for lead, trail, par.
        The shared SAi of `ri` is rewritten in place."""
        kids, structs = [], []
        for cnt in (nl, nt, np_):
            c = self.RevealIndex()
            if cnt > 0:
                sa = np.zeros(cnt, dtype=self.sa_t)
                lcp = np.zeros(cnt, dtype=self.lcp_t)
                c.SA, c.LCP, c.n = sa.ctypes.data, lcp.ctypes.data, cnt
                kids.append((sa, lcp))
            else:
                kids.append(None)
            structs.append(c)
        D = np.ascontiguousarray(D, dtype=np.uint8)
        self.lib.split(ctypes.byref(ri), D.ctypes.data, *[ctypes.byref(s) for s in structs])
        return kids

    def bubble_sort(self, tbuf, SA, LCP, SAi, matching):
        """reference bubble_sort() on a child (arrays modified in place)."""
        c = self.RevealIndex()
        c.T = tbuf.ctypes.data
        c.SA, c.LCP, c.SAi, c.n = SA.ctypes.data, LCP.ctypes.data, SAi.ctypes.data, len(SA)
        self.lib.bubble_sort(ctypes.byref(c), [(int(b), int(e)) for b, e in matching])

    def extract(self, tbuf, SA, LCP, SAi, nsep, intervals, rc=0, nT=None):
        """reference extract() (reveal.c:1386-1505).  It frees idx->SA / idx->LCP and
        installs malloc'ed replacements, so the struct gets malloc'ed copies.  tbuf and
        SAi are modified in place.  -> (SA, LCP, intervals) with SA[0] = whatever the
        heap held (the reference never writes it)."""
        libc = ctypes.CDLL(None)
        libc.malloc.restype = ctypes.c_void_p
        libc.malloc.argtypes = [ctypes.c_size_t]
        libc.free.argtypes = [ctypes.c_void_p]
        c = self.RevealIndex()
        c.ob_refcnt = 1 << 30
        sa_p = libc.malloc(max(SA.nbytes, 8)); lcp_p = libc.malloc(max(LCP.nbytes, 8))
        ctypes.memmove(sa_p, SA.ctypes.data, SA.nbytes); ctypes.memmove(lcp_p, LCP.ctypes.data, LCP.nbytes)
        nsep = np.ascontiguousarray(np.asarray(nsep, dtype=self.sa_t))
        c.T, c.SA, c.LCP, c.SAi, c.n = tbuf.ctypes.data, sa_p, lcp_p, SAi.ctypes.data, len(SA)
        c.nT = nT if nT is not None else len(SA)
        c.nsep, c.rc = nsep.ctypes.data, rc
        iv = [(int(b), int(e)) for b, e in intervals]
        self.lib.extract(ctypes.byref(c), (iv,), None)
        nn = int(c.n)
        oSA = np.ctypeslib.as_array(ctypes.cast(c.SA, ctypes.POINTER(self.c_sa)), shape=(max(nn, 1),))[:nn].astype(self.sa_t)
        oLCP = np.ctypeslib.as_array(ctypes.cast(c.LCP, ctypes.POINTER(self.c_lcp)), shape=(max(nn, 1),))[:nn].astype(self.lcp_t)
        libc.free(c.SA); libc.free(c.LCP)
        return oSA, oLCP, iv
