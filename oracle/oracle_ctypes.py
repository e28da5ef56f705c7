"""ctypes binding of oracle/liboracle[64].so -- TEST INFRASTRUCTURE ONLY.

The CPU restatement of the reveallib hot path (oracle/reveal_oracle.c).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this;
the product (reveal_amd/) never does.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """compile liboracle.so / liboracle64.so (gcc, seconds)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, f)) for f in ("liboracle.so", "liboracle64.so"))
    if not need:
        src = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("reveal_oracle.c", "reveal_oracle.h"))
        need = any(os.path.getmtime(os.path.join(_HERE, f)) < src for f in ("liboracle.so", "liboracle64.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Oracle:
    def __init__(self, sa64=False, use_ref_divsufsort=True):
        build()
        self.sa64 = sa64
        self.sa_t = np.int64 if sa64 else np.int32
        self.lcp_t = np.uint32 if sa64 else np.int32
        self.c_sa = ctypes.c_int64 if sa64 else ctypes.c_int32
        self.c_lcp = ctypes.c_uint32 if sa64 else ctypes.c_int32
        self.lib = L = ctypes.CDLL(os.path.join(_HERE, "liboracle64.so" if sa64 else "liboracle.so"))
        V = ctypes.c_void_p
        c_sa = self.c_sa
        L.ro_suffix_array.argtypes = [V, V, c_sa]
        L.ro_suffix_array_own.argtypes = [V, V, c_sa]
        L.ro_sufcheck.argtypes = [V, V, c_sa]
        L.ro_set_divsufsort.argtypes = [V]
        L.ro_inverse.argtypes = [V, V, c_sa]
        L.ro_compute_lcp.argtypes = [V, V, V, V, c_sa]
        L.ro_build_so.argtypes = [V, V, ctypes.c_int, c_sa]
        L.ro_revcomp.argtypes = [V, c_sa]
        L.ro_getmums.restype = ctypes.c_int64
        L.ro_getmums.argtypes = [V, ctypes.c_int, ctypes.c_int, V, V, V, ctypes.c_int64]
        L.ro_getmultimums.restype = ctypes.c_int64
        L.ro_getmultimums.argtypes = [V, ctypes.c_int, ctypes.c_int, ctypes.c_int, V, V, V, ctypes.c_int64,
                                      V, V, ctypes.c_int64, V]
        L.ro_label.argtypes = [V, c_sa, V, V, ctypes.c_int, V, ctypes.c_int, V, ctypes.c_int, V, ctypes.c_int,
                               self.c_lcp, V, V, V]
        L.ro_split.argtypes = [V, V, c_sa, V, V] + [V] * 6 + [V] * 3
        L.ro_bubble_sort.argtypes = [V, V, c_sa, V, V, ctypes.c_int]
        L.ro_splitindex.argtypes = [V, V, V, c_sa, V, V, V, ctypes.c_int] + [V, ctypes.c_int] * 4 + [V] * 6 + [V, V]
        L.ro_splitindex.restype = None
        L.ro_extract.argtypes = [V, V, V, c_sa, V, V, c_sa, ctypes.c_int, V, ctypes.c_int, V, V]
        L.ro_extract.restype = c_sa
        L.ro_align.argtypes = [V, V, V, c_sa, V, ctypes.c_int, V, V, V, ctypes.c_int, ctypes.c_int,
                               V, ctypes.c_int64, V]
        L.ro_align.restype = ctypes.c_int
        self.ref_divsufsort = False
        if use_ref_divsufsort:
            try:
                from . import ref_ctypes
            except ImportError:  # run as a script
                import ref_ctypes
            if ref_ctypes.available(sa64):
                self._ref = ref_ctypes.Ref(sa64)          # keep the library alive
                L.ro_set_divsufsort(self._ref.divsufsort_addr())
                self.ref_divsufsort = True

        class View(ctypes.Structure):
            _fields_ = [("T", V), ("SA", V), ("LCP", V), ("SO", V), ("nsep", V),
                        ("n", c_sa), ("nT", c_sa), ("main_nsamples", ctypes.c_int), ("rc", ctypes.c_int)]
        self.View = View

        class Main(ctypes.Structure):
            _fields_ = [("T", V), ("SAi", V), ("SO", V), ("nsep", V), ("nT", c_sa), ("nsamples", ctypes.c_int)]
        self.Main = Main

        class Stats(ctypes.Structure):
            _fields_ = [("nsteps", ctypes.c_int64), ("nsplits", ctypes.c_int64), ("anchored_bp", ctypes.c_int64),
                        ("maxdepth", ctypes.c_int), ("t_scan", ctypes.c_double), ("t_pick", ctypes.c_double),
                        ("t_split", ctypes.c_double), ("t_bubble", ctypes.c_double)]
        self.Stats = Stats

        class BenchCtx(ctypes.Structure):
            _fields_ = [("nanchors", ctypes.c_int64), ("cap_anchors", ctypes.c_int64), ("l", V), ("n", V), ("off", V),
                        ("npos", ctypes.c_int64), ("cap_pos", ctypes.c_int64), ("pos", V)]
        self.BenchCtx = BenchCtx
        sa, lcp = self.sa_t, self.lcp_t
        self.trace_dtype = np.dtype([("key", sa), ("n", sa), ("depth", np.int32), ("nsamples", np.int32),
                                     ("nnodes", np.int32), ("nmums", np.int64), ("picked", np.int32), ("l", lcp),
                                     ("mn", np.int32), ("sp_min", sa), ("h_sa", np.uint64), ("h_lcp", np.uint64),
                                     ("h_mums", np.uint64)], align=True)

    # -- construct pieces -----------------------------------------------------
    @staticmethod
    def textbuf(T):
        t = np.zeros(len(T) + 1, dtype=np.uint8)
        t[:len(T)] = np.frombuffer(bytes(T), dtype=np.uint8) if not isinstance(T, np.ndarray) else T[:len(T)]
        return t

    def suffix_array(self, tbuf, n=None, own=False):
        n = len(tbuf) - 1 if n is None else n
        SA = np.empty(n, dtype=self.sa_t)
        fn = self.lib.ro_suffix_array_own if own else self.lib.ro_suffix_array
        if fn(tbuf.ctypes.data, SA.ctypes.data, n) != 0:
            raise RuntimeError("suffix array construction failed")
        return SA

    def sufcheck(self, tbuf, SA):
        return self.lib.ro_sufcheck(tbuf.ctypes.data, SA.ctypes.data, len(SA))

    def inverse(self, SA):
        SAi = np.empty_like(SA)
        self.lib.ro_inverse(SA.ctypes.data, SAi.ctypes.data, len(SA))
        return SAi

    def compute_lcp(self, tbuf, SA, SAi):
        LCP = np.zeros(len(SA), dtype=self.lcp_t)
        self.lib.ro_compute_lcp(tbuf.ctypes.data, SA.ctypes.data, SAi.ctypes.data, LCP.ctypes.data, len(SA))
        return LCP

    def build_so(self, nsep, nsamples, n):
        SO = np.zeros(n, dtype=np.uint16)
        nsep = np.ascontiguousarray(np.asarray(nsep, dtype=self.sa_t))
        self.lib.ro_build_so(SO.ctypes.data, nsep.ctypes.data, nsamples, n)
        return SO

    def revcomp(self, buf):
        self.lib.ro_revcomp(buf.ctypes.data, len(buf))

    def construct(self, T, nsep, nsamples):
        """T bytes -> dict(tbuf, SA, SAi, LCP, SO)   (interface.c:160-291)"""
        tbuf = self.textbuf(T)
        SA = self.suffix_array(tbuf)
        SAi = self.inverse(SA)
        LCP = self.compute_lcp(tbuf, SA, SAi)
        SO = self.build_so(nsep, nsamples, len(SA)) if nsamples > 2 else None
        return dict(tbuf=tbuf, SA=SA, SAi=SAi, LCP=LCP, SO=SO,
                    nsep=np.ascontiguousarray(np.asarray(nsep, dtype=self.sa_t)), nsamples=nsamples)

    # -- scans ----------------------------------------------------------------
    def _view(self, tbuf, SA, LCP, nsep, main_nsamples, SO=None, nT=None, rc=0):
        v = self.View()
        nsep = np.ascontiguousarray(np.asarray(nsep, dtype=self.sa_t))
        v.T, v.SA, v.LCP, v.SO, v.nsep = tbuf.ctypes.data, SA.ctypes.data, LCP.ctypes.data, _ptr(SO), nsep.ctypes.data
        v.n = len(SA)
        v.nT = nT if nT is not None else len(SA)
        v.main_nsamples = main_nsamples
        v.rc = rc
        v._keep = (tbuf, SA, LCP, SO, nsep)
        return v

    def getmums(self, tbuf, SA, LCP, nsep, minl, rem=False, rc=0, nT=None):
        """-> (l, a, b) arrays in rank order  (reveal.c:55-116 / :119-180)"""
        v = self._view(tbuf, SA, LCP, nsep, 2, nT=nT, rc=rc)
        cap = 1024
        while True:
            l = np.empty(cap, dtype=self.lcp_t); a = np.empty(cap, dtype=self.sa_t); b = np.empty(cap, dtype=self.sa_t)
            r = self.lib.ro_getmums(ctypes.byref(v), minl, 1 if rem else 0, l.ctypes.data, a.ctypes.data, b.ctypes.data, cap)
            if r <= cap:
                return l[:r].copy(), a[:r].copy(), b[:r].copy()
            cap = r

    def getmultimums(self, tbuf, SA, LCP, SO, nsep, main_nsamples, minl=0, minn=2, mems=False):
        """-> (l, n, off, so, pos) CSR in emission order  (reveal.c:436-580 / :292-434)"""
        v = self._view(tbuf, SA, LCP, nsep, main_nsamples, SO=SO)
        capm, capp = 1024, 4096
        while True:
            l = np.empty(capm, dtype=self.lcp_t); n = np.empty(capm, dtype=np.int32)
            off = np.zeros(capm + 1, dtype=np.int64)
            so = np.empty(capp, dtype=np.uint16); pos = np.empty(capp, dtype=self.sa_t)
            need = ctypes.c_int64(0)
            r = self.lib.ro_getmultimums(ctypes.byref(v), minl, minn, 1 if mems else 0, l.ctypes.data, n.ctypes.data,
                                         off.ctypes.data, capm, so.ctypes.data, pos.ctypes.data, capp, ctypes.byref(need))
            if r >= 0:
                return l[:r].copy(), n[:r].copy(), off[:r + 1].copy(), so[:need.value].copy(), pos[:need.value].copy()
            capm, capp = -r - 1 + 16, need.value + 16

    # -- split / bubble -------------------------------------------------------
    def _iv(self, iv):
        a = np.ascontiguousarray(np.asarray(iv, dtype=self.sa_t).reshape(-1, 2))
        return a, len(a)

    def label(self, n, SAi, lead, trail, rest, sp, l):
        D = np.zeros(max(n, 1), dtype=np.uint8)
        la, nl_ = self._iv(lead); ta, nt_ = self._iv(trail); ra, nr_ = self._iv(rest)
        spa = np.ascontiguousarray(np.asarray(sp, dtype=self.sa_t))
        c = [self.c_sa(0), self.c_sa(0), self.c_sa(0)]
        self.lib.ro_label(D.ctypes.data, n, SAi.ctypes.data, la.ctypes.data, nl_, ta.ctypes.data, nt_, ra.ctypes.data, nr_,
                          spa.ctypes.data, len(spa), l, ctypes.byref(c[0]), ctypes.byref(c[1]), ctypes.byref(c[2]))
        return D[:n], c[0].value, c[1].value, c[2].value

    def split(self, SA, LCP, D, SAi, nl, nt, np_):
        kids = []
        for cnt in (nl, nt, np_):
            kids.append((np.zeros(max(cnt, 1), dtype=self.sa_t), np.zeros(max(cnt, 1), dtype=self.lcp_t)))
        o = [self.c_sa(0), self.c_sa(0), self.c_sa(0)]
        D = np.ascontiguousarray(D, dtype=np.uint8)
        self.lib.ro_split(SA.ctypes.data, LCP.ctypes.data, len(SA), D.ctypes.data, SAi.ctypes.data,
                          kids[0][0].ctypes.data, kids[0][1].ctypes.data, kids[1][0].ctypes.data, kids[1][1].ctypes.data,
                          kids[2][0].ctypes.data, kids[2][1].ctypes.data,
                          ctypes.byref(o[0]), ctypes.byref(o[1]), ctypes.byref(o[2]))
        return [(k[0][:c.value], k[1][:c.value]) if c.value > 0 else None for k, c in zip(kids, o)]

    def bubble_sort(self, SA, LCP, SAi, match_begins):
        mb = np.ascontiguousarray(np.asarray(match_begins, dtype=self.sa_t))
        self.lib.ro_bubble_sort(SA.ctypes.data, LCP.ctypes.data, len(SA), SAi.ctypes.data, mb.ctypes.data, len(mb))

    # -- host-driven single steps (reveal.c:1386-1748) -----------------------------
    def splitindex(self, tbuf, SA, LCP, SAi, SO, nsep, main_nsamples, lead, trail, match, rest):
        """ro_splitindex: tbuf is lower-cased and SAi rewritten in place.
        -> [(SA, LCP, nsamples) | None] * 3 for the leading, trailing and parallel child"""
        la, nl_ = self._iv(lead); ta, nt_ = self._iv(trail); ma, nm_ = self._iv(match); ra, nr_ = self._iv(rest)
        sizes = [int((x[:, 1] - x[:, 0]).sum()) if len(x) else 0 for x in (la, ta, ra)]
        kids = [(np.zeros(max(c, 1), dtype=self.sa_t), np.zeros(max(c, 1), dtype=self.lcp_t)) for c in sizes]
        counts = np.zeros(3, dtype=self.sa_t); ns = np.zeros(3, dtype=np.int32)
        nsep = np.ascontiguousarray(np.asarray(nsep, dtype=self.sa_t))
        self.lib.ro_splitindex(tbuf.ctypes.data, SA.ctypes.data, LCP.ctypes.data, len(SA), SAi.ctypes.data, _ptr(SO), nsep.ctypes.data,
                               main_nsamples, la.ctypes.data, nl_, ta.ctypes.data, nt_, ma.ctypes.data, nm_, ra.ctypes.data, nr_,
                               kids[0][0].ctypes.data, kids[0][1].ctypes.data, kids[1][0].ctypes.data, kids[1][1].ctypes.data,
                               kids[2][0].ctypes.data, kids[2][1].ctypes.data, counts.ctypes.data, ns.ctypes.data)
        return [(k[0][:c], k[1][:c], int(q)) if c > 0 else None for k, c, q in zip(kids, counts.tolist(), ns)]

    def extract(self, tbuf, SA, LCP, SAi, nsep, intervals, rc=0, nT=None):
        """ro_extract: tbuf lower-cased, SAi rewritten in place -> (SA, LCP, intervals after the rc remap)"""
        iv, niv = self._iv(intervals)
        iv = iv.copy()
        oSA = np.zeros(max(len(SA), 1), dtype=self.sa_t); oLCP = np.zeros(max(len(SA), 1), dtype=self.lcp_t)
        nsep = np.ascontiguousarray(np.asarray(nsep, dtype=self.sa_t))
        nn = self.lib.ro_extract(tbuf.ctypes.data, SA.ctypes.data, LCP.ctypes.data, len(SA), SAi.ctypes.data, nsep.ctypes.data,
                                 nT if nT is not None else len(SA), rc, iv.ctypes.data, niv, oSA.ctypes.data, oLCP.ctypes.data)
        if nn < 0:
            raise ValueError("extract: rank 0 is matched (the reference overruns its buffers there)")
        return oSA[:nn].copy(), oLCP[:nn].copy(), [(int(b), int(e)) for b, e in iv]

    # -- the recursion with the bench callbacks --------------------------------
    def align_bench(self, cons, nodes, minl, minn=2, trace_cap=0, anchor_cap=None):
        """runs ro_align with ro_bench_picker / ro_bench_graphalign.
        cons: dict from construct() (SA/LCP are consumed; T is lower-cased in place).
        -> dict(trace, anchors=(l, n, off, pos), stats, T)"""
        n = len(cons["SA"])
        m = self.Main()
        m.T, m.SAi, m.SO, m.nsep = cons["tbuf"].ctypes.data, cons["SAi"].ctypes.data, _ptr(cons["SO"]), cons["nsep"].ctypes.data
        m.nT, m.nsamples = n, cons["nsamples"]
        # ro_align frees SA/LCP with free(): hand it malloc'ed copies
        libc = ctypes.CDLL(None)
        libc.malloc.restype = ctypes.c_void_p
        libc.malloc.argtypes = [ctypes.c_size_t]
        sa_p = libc.malloc(max(cons["SA"].nbytes, 8)); lcp_p = libc.malloc(max(cons["LCP"].nbytes, 8))
        ctypes.memmove(sa_p, cons["SA"].ctypes.data, cons["SA"].nbytes)
        ctypes.memmove(lcp_p, cons["LCP"].ctypes.data, cons["LCP"].nbytes)
        nodes_a, nn = self._iv(nodes)
        trace = np.zeros(max(trace_cap, 1), dtype=self.trace_dtype)
        cap_a = anchor_cap if anchor_cap is not None else max(16, n // 8)
        cap_p = cap_a * max(2, cons["nsamples"])
        al = np.zeros(cap_a, dtype=self.lcp_t); an = np.zeros(cap_a, dtype=np.int32)
        aoff = np.zeros(cap_a + 1, dtype=np.int64); apos = np.zeros(cap_p, dtype=self.sa_t)
        ctx = self.BenchCtx(0, cap_a, al.ctypes.data, an.ctypes.data, aoff.ctypes.data, 0, cap_p, apos.ctypes.data)
        st = self.Stats()
        picker = ctypes.cast(self.lib.ro_bench_picker, ctypes.c_void_p)
        galign = ctypes.cast(self.lib.ro_bench_graphalign, ctypes.c_void_p)
        r = self.lib.ro_align(ctypes.byref(m), sa_p, lcp_p, n, nodes_a.ctypes.data, nn, picker, galign,
                              ctypes.cast(ctypes.byref(ctx), ctypes.c_void_p), minl, minn,
                              trace.ctypes.data if trace_cap else None, trace_cap, ctypes.byref(st))
        if r != 0:
            raise RuntimeError("ro_align failed")
        if ctx.nanchors > cap_a:
            raise RuntimeError("anchor buffer too small: %d > %d" % (ctx.nanchors, cap_a))
        na = ctx.nanchors
        stats = {f[0]: getattr(st, f[0]) for f in self.Stats._fields_}
        return dict(trace=trace[:min(trace_cap, st.nsteps)].copy(),
                    anchors=(al[:na].copy(), an[:na].copy(), aoff[:na + 1].copy(), apos[:ctx.npos].copy()),
                    stats=stats, T=bytes(cons["tbuf"][:n]))
