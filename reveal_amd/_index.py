"""The `index` type of reveallib / reveallib64, on top of the HIP C ABI.

Host-side mirror of the reference's CPython type (reveallib/interface.c:474-487
method table, :731-785 getset table): same method names, argument meaning and
error behaviour, so reveal/rem.py-style callers run unchanged:

    idx = reveallib.index(sa="", lcp="", cache=0)
    idx.addsample(name); idx.addsequence(seq) -> (begin, end)
    idx.construct(rc=0)
    idx.getmums(minl); idx.getmultimums(minlength=0, minn=2); idx.getmultimems(...)
    idx.align(mumpicker, align, threads=0, wpen=0, wscore=0, minl=0, minn=0)
    idx.n .depth .nsamples .samples .nodes .leftnode .rightnode .nsep .SA .SAi .SO .LCP .T

All compute happens in HBM through libreveal_amd[64].so; nothing here falls
back to the CPU.
"""
import ctypes
import mmap
import os
import sys
import numpy as np

from . import _lib
from ._lib import RV_T, RV_SA, RV_SAI, RV_LCP, RV_SO, RV_NSEP


def _page_array(count, dtype):
    """an array on pages of its own (anonymous mmap, whole pages): what rv_set_result_buffers page-locks must not share a page with
    anything else -- arrays from the C heap do (with one another, with whatever numpy allocates next), and locking / unlocking a range
    acts on whole pages: a later copy into a neighbour of a released array ended in a GPU memory fault now and then"""
    nbytes = max(int(count), 1) * np.dtype(dtype).itemsize
    m = mmap.mmap(-1, (nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE)
    return np.frombuffer(m, dtype=dtype, count=max(int(count), 1))


def _pairs(iv, sa64):
    """iterable of (begin, end) -> contiguous int64 array + count (iteration order kept)"""
    a = np.ascontiguousarray(np.array([(int(b), int(e)) for b, e in iv], dtype=np.int64).reshape(-1, 2))
    return a, len(a)


_OPTION_DEFAULTS = {}      # (library width, switch) -> value a fresh handle carries


def make_index_type(sa64, error):
    """class factory: one `index` type per module (reveallib, reveallib64)."""

    class index(object):
        """Reveal Index (interface.c:841-881)"""

        def __init__(self, sa="", lcp="", cache=0):       # reveal_init, interface.c:489-521
            self._lib = _lib.get(sa64)
            self._dll = self._lib.dll
            self._h = None
            self._safile, self._lcpfile, self._cache = sa or "", lcp or "", int(cache)
            self._samples, self._nodes, self.skipmums = [], set(), []
            self._leftnode = self._rightnode = None
            self._depth = 0
            self._constructed = False
            self._main = self
            self._slot = None            # frontier slot while align() runs (sub-indices)
            self._sx = None              # detached (sub)index behind splitindex / extract (include/reveal_amd.h, rv_sx_*)
            self._n_sub = None
            self._nsamples_sub = None
            self._h = self._dll.rv_new(_lib.device())
            if not self._h:
                raise error(self._lib.err())
            self.options_from_env()

        # ---- switches (not in the reference) ------------------------------------
        def set_option(self, name, value=1):
            """one switch of this handle (include/reveal_amd.h rv_set_option; names = the RV_* spellings of rv_common.h RV_OPTION_LIST)"""
            if self._dll.rv_set_option(self._main._h, name.encode(), int(value)) != 0:
                self._fail()

        def get_option(self, name):
            v = ctypes.c_int64(0)
            if self._dll.rv_get_option(self._main._h, name.encode(), ctypes.byref(v)) != 0:
                self._fail()
            return v.value

        def options_from_env(self):
            """The library never reads the environment; this layer does, when a handle is made (and again when asked to): every RV_*
            variable that names a switch is applied (empty value = 1), every switch without a variable goes back to its default."""
            for k in range(self._dll.rv_option_count()):
                name = self._dll.rv_option_name(k).decode()
                v = os.environ.get(name)
                if v is None:
                    d = _OPTION_DEFAULTS.get((sa64, name))
                    if d is None:
                        d = _OPTION_DEFAULTS[(sa64, name)] = self.get_option(name)      # (a fresh handle: what rv_new set)
                    elif self.get_option(name) != d:
                        self.set_option(name, d)
                    continue
                try:
                    iv = int(v) if v.strip() else 1
                except ValueError:
                    iv = 1
                _OPTION_DEFAULTS.setdefault((sa64, name), self.get_option(name))
                self.set_option(name, iv)

        # the main index of a sub-index (itself for a main index) -- kept as "None = self": an attribute that points back at its own object is
        # a reference cycle, and the handle (streams, pinned buffers, the index in HBM) would only be released when the cycle collector runs
        @property
        def _main(self):
            m = self.__dict__.get("_main_ref")
            return self if m is None else m

        @_main.setter
        def _main(self, v):
            self.__dict__["_main_ref"] = None if v is self else v

        def __del__(self):
            try:
                if getattr(self, "_sx", None):
                    self._dll.rv_sx_free(self._sx)     # (children keep their main object alive through _main)
                    self._sx = None
                if self._h and self._main is self:
                    self._dll.rv_free(self._h)
                    self._h = None
            except Exception:
                pass

        def _fail(self, exc=None):
            raise (exc or error)(self._lib.err())

        # ---- text assembly ---------------------------------------------------
        def addsample(self, *args):                         # interface.c:18-49
            if len(args) < 1:
                raise error("Specify name of sample as argument.")
            if not isinstance(args[0], str):
                raise error("Sample name has to be a string.")
            self._samples.append(args[0])
            self._dll.rv_add_sample(self._h)
            self._constructed = False          # (the arrays in HBM describe the text as it was)
            return None

        def addsequence(self, seq):                         # interface.c:51-95
            if isinstance(seq, str):
                seq = seq.encode()
            elif not isinstance(seq, (bytes, bytearray)):
                raise TypeError("argument 1 must be str or bytes")
            b, e = ctypes.c_int64(0), ctypes.c_int64(0)
            if self._dll.rv_add_sequence(self._h, bytes(seq), len(seq), ctypes.byref(b), ctypes.byref(e)) != 0:
                self._fail()
            intv = (b.value, e.value)
            self._nodes.add(intv)
            self._constructed = False
            return intv

        def reset(self, reserve=0):
            """Not in the reference (its callers make a new index per input): forget the text and the samples, keep the handle's
            device arrays, SA-build scratch, streams and (page-locked) host text for the next input (rv_reset); `reserve` = bytes
            of text to come (sequences + one separator each), so the host buffer is made once (rv_reserve_text)."""
            if self._main is not self or self._slot is not None:
                raise error("reset() is for a main index outside align()")
            if self._sx:
                self._dll.rv_sx_free(self._sx)
                self._sx = None
            if self._dll.rv_reset(self._h) != 0:
                self._fail()
            if reserve and self._dll.rv_reserve_text(self._h, int(reserve)) != 0:
                self._fail()
            self._samples, self._nodes, self.skipmums = [], set(), []
            self._leftnode = self._rightnode = None
            self._depth = 0
            self._constructed = False
            self._pending = False
            self._n_sub = self._nsamples_sub = None

        def upload(self):
            """Not in the reference: copy the assembled text to HBM now (construct()
            does it on demand), so a timed construct() starts from resident input."""
            if self._dll.rv_upload(self._h) != 0:
                self._fail()

        def upload_again(self):
            """the host->device copy of the text once more (rv_upload_again): its cost without the first call's allocations"""
            if self._dll.rv_upload_again(self._h) != 0:
                self._fail()

        def prof(self, enable=None, reset=False, only=None):
            """HIP-event kernel timing on the index' stream -> {kernel: (launches, ms, bytes)}
            only = names of the kernel classes to time (default: all; every timed span costs the stream two events)"""
            ids = {"scan_pair": _lib.K_SCAN_PAIR, "scan_multi": _lib.K_SCAN_MULTI, "sa_build": _lib.K_SA_SORT, "lcp": _lib.K_LCP,
                   "split": _lib.K_SPLIT, "label": _lib.K_LABEL, "bubble": _lib.K_BUBBLE, "radix_scatter": _lib.K_RADIX_SCATTER,
                   "radix_hist": _lib.K_RADIX_HIST, "text_round": _lib.K_TEXT_ROUND, "cascade": _lib.K_CASCADE, "diag_table": _lib.K_DIAG_TABLE,
                   "init_keys": _lib.K_INIT_KEYS, "publish": _lib.K_PUBLISH}
            if enable is not None:
                on = 0
                if enable:
                    on = 1 if only is None else sum(2 << ids[k] for k in only)
                self._dll.rv_prof_enable(self._h, on)
            if reset:
                self._dll.rv_prof_reset(self._h)
            out = {}
            for name, k in ids.items():
                n, ms, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
                self._dll.rv_prof_get(self._h, k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by))
                out[name] = (n.value, ms.value, by.value)
            return out

        def sa_stats(self):
            v = [ctypes.c_int(0) for _ in range(4)]
            se, rp = ctypes.c_int64(0), ctypes.c_int(0)
            self._dll.rv_sa_stats(self._h, *[ctypes.byref(x) for x in v], ctypes.byref(se), ctypes.byref(rp))
            tail = (ctypes.c_int64 * 2)()
            self._dll.rv_sa_tail(self._h, tail)
            return dict(sigma=v[0].value, bits=v[1].value, k0=v[2].value, rounds=v[3].value, sorted_elems=se.value, radix_passes=rp.value,
                        diag_table=int(self._dll.rv_sa_diag_table(self._h)), far_pairs=int(tail[0]), lcp_list=int(tail[1]))

        # ---- construct --------------------------------------------------------
        def construct(self, rc=0):                          # interface.c:160-291
            if self._sx:
                self._dll.rv_sx_free(self._sx)
                self._sx = None
            r = self._dll.rv_construct(self._h, int(rc), self._safile.encode(), self._lcpfile.encode(), self._cache)
            if r != 0:
                self._fail()
            self._constructed = True
            self._rc = 1 if int(rc) == 1 else 0
            self._depth = 0
            self._main = self
            return None

        # ---- getters (interface.c:538-729) -----------------------------------------
        def _sx_info(self):
            info = _lib.RvSub()
            self._dll.rv_sx_info(self._sx, ctypes.byref(info))
            return info

        @property
        def n(self):
            if self._sx:
                return self._sx_info().n
            return self._n_sub if self._n_sub is not None else self._dll.rv_n(self._h)

        @property
        def depth(self):
            return self._depth

        @property
        def nsamples(self):
            if self._sx and self._depth > 0:
                return self._sx_info().nsamples
            return self._nsamples_sub if self._nsamples_sub is not None else self._dll.rv_nsamples(self._h)

        @property
        def samples(self):
            return self._main._samples

        @property
        def nodes(self):
            if self.__dict__.get("_nodes_stale"):      # (sequences were added inside the library: rv_graph_read_gfa -- 10^7 tuples are made when somebody asks)
                self._sync_nodes()
            return self._nodes

        @property
        def leftnode(self):
            return self._leftnode

        @property
        def rightnode(self):
            return self._rightnode

        @property
        def nsep(self):
            ns = self._dll.rv_nsamples(self._h)
            out = np.zeros(max(ns - 1, 1), dtype=np.int64)
            k = self._dll.rv_get_array(self._h, RV_NSEP, out.ctypes.data, len(out))
            return [int(x) for x in out[:max(k, 0)]]

        def _array(self, which, dtype, count, exc=TypeError):
            out = np.zeros(max(count, 1), dtype=dtype)
            if self._sx and which in (RV_SA, RV_LCP):
                k = self._dll.rv_sx_array(self._sx, which, out.ctypes.data, len(out))
            elif self._slot is not None and which in (RV_SA, RV_LCP):
                k = self._dll.rv_sub_array(self._h, self._slot, which, out.ctypes.data, len(out))
            else:
                k = self._dll.rv_get_array(self._h, which, out.ctypes.data, len(out))
            if k < 0:
                raise exc(self._lib.err())
            return out[:k]

        def _main_n(self):
            return self._dll.rv_n(self._h)

        @property
        def SA(self):
            return self._array(RV_SA, self._lib.sa_t, self.n).tolist()

        @property
        def LCP(self):
            return self._array(RV_LCP, self._lib.lcp_t, self.n).tolist()

        @property
        def SAi(self):
            return self._array(RV_SAI, self._lib.sa_t, self._main_n()).tolist()

        @property
        def SO(self):
            return self._array(RV_SO, np.uint16, self._main_n()).tolist()

        @property
        def T(self):
            return self._array(RV_T, np.uint8, self._main_n(), exc=error).tobytes().decode("latin-1")

        # numpy views of the same (not in the reference; cheaper than lists)
        def array(self, name):
            which = {"SA": RV_SA, "LCP": RV_LCP, "SAi": RV_SAI, "SO": RV_SO, "T": RV_T}[name]
            dt = {RV_SA: self._lib.sa_t, RV_LCP: self._lib.lcp_t, RV_SAI: self._lib.sa_t, RV_SO: np.uint16, RV_T: np.uint8}[which]
            return self._array(which, dt, self.n if which in (RV_SA, RV_LCP) else self._main_n())

        # ---- scans --------------------------------------------------------------
        def _sx_mums(self, minl, minn):
            """scan of a detached (sub)index -> [(l, n, ((sample, pos), ...))]"""
            mem = ctypes.c_int64(0)
            cnt = self._dll.rv_sx_scan(self._sx, int(minl), int(minn), ctypes.byref(mem))
            if cnt < 0:
                self._fail()
            l = np.zeros(max(cnt, 1), dtype=np.uint32); n = np.zeros(max(cnt, 1), dtype=np.int32)
            off = np.zeros(cnt + 1, dtype=np.int64)
            so = np.zeros(max(mem.value, 1), dtype=np.uint16); pos = np.zeros(max(mem.value, 1), dtype=np.int64)
            if self._dll.rv_sx_fetch(self._sx, l.ctypes.data, n.ctypes.data, off.ctypes.data, so.ctypes.data, pos.ctypes.data) != 0:
                self._fail()
            return _csr_to_tuples(cnt, l, n, off, so, pos)

        def getmums(self, minl=0):                          # reveal.c:55-116
            if self._sx:
                if self._dll.rv_nsamples(self._h) != 2:
                    raise error("getmums on a sub-index needs a two-sample main index")
                rc = 1 if getattr(self._main, "_rc", 0) else 0
                out = [(l, (spd[0][1], spd[1][1]), rc) for l, _, spd in self._sx_mums(minl, 0)]
                if rc:                                                            # reveal.c:98-100
                    nsep0, nT = self.nsep[0], self._main_n()
                    out = [(l, (a, nsep0 + (nT - b - l)), rc) for l, (a, b), _ in out]
                return out
            cnt = self._dll.rv_getmums(self._h, int(minl))
            if cnt < 0:
                self._fail(TypeError if cnt == -2 else error)
            l = np.zeros(max(cnt, 1), dtype=np.uint32)
            a = np.zeros(max(cnt, 1), dtype=np.int64)
            b = np.zeros(max(cnt, 1), dtype=np.int64)
            if self._dll.rv_fetch_mums(self._h, l.ctypes.data, a.ctypes.data, b.ctypes.data, len(l)) != 0:
                self._fail()
            rc = 1 if getattr(self, "_rc", 0) else 0
            return [(int(l[k]), (int(a[k]), int(b[k])), rc) for k in range(cnt)]

        def _multi(self, minlength, minn, mems):
            if self._sx:
                if mems or self._dll.rv_nsamples(self._h) <= 2:
                    raise error("only getmultimums of an index with more than two samples is available on a sub-index")
                return self._sx_mums(minlength, minn)
            members = ctypes.c_int64(0)
            cnt = self._dll.rv_getmultimums(self._h, int(minlength), int(minn), mems, ctypes.byref(members))
            if cnt < 0:
                self._fail(TypeError if cnt == -2 else error)
            l = np.zeros(max(cnt, 1), dtype=np.uint32); n = np.zeros(max(cnt, 1), dtype=np.int32)
            off = np.zeros(cnt + 1, dtype=np.int64)
            so = np.zeros(max(members.value, 1), dtype=np.uint16); pos = np.zeros(max(members.value, 1), dtype=np.int64)
            if self._dll.rv_fetch_multi(self._h, l.ctypes.data, n.ctypes.data, off.ctypes.data, so.ctypes.data, pos.ctypes.data) != 0:
                self._fail()
            return _csr_to_tuples(cnt, l, n, off, so, pos)

        def getmultimums(self, minlength=0, minn=2):        # reveal.c:436-580
            return self._multi(minlength, minn, 0)

        def getmultimems(self, minlength=0, minn=2):        # reveal.c:292-434
            return self._multi(minlength, minn, 1)

        # ---- the recursion -----------------------------------------------------------
        def align(self, mumpicker, align, threads=0, wpen=0, wscore=0, minl=0, minn=0):
            """interface.c:293-415 + aligner() reveal.c:731-1338.

            Same callback contracts as the reference; sub-indices are visited
            level by level instead of LIFO (children of a split are independent,
            the anchor set is the same).

            `threads` is accepted for call compatibility and ignored: the reference's worker threads (interface.c:338-386)
            only run its C work concurrently -- here every sub-index of a level is processed by the same kernel launches, and
            the Python callbacks run on the calling thread one after the other, as they do under the reference's `python`
            mutex + GIL (reveal.c:779-780).  `wpen` / `wscore` are stored by the reference and never read in C; ignored too."""
            if not self._constructed:
                raise error("Index not yet constructed, alignment stopped.")
            dll, h = self._dll, self._h
            if dll.rv_align_begin(h, int(minl), int(minn)) != 0:
                self._fail()
            self._depth = 0
            self._slot = 0
            frontier = [self]
            try:
                while frontier:
                    if dll.rv_frontier_scan(h) != 0:
                        self._fail()
                    decided = {}
                    for s, idx in enumerate(frontier):
                        if not callable(mumpicker):
                            raise TypeError("**** mumpicker isn't callable")      # reveal.c:783-792
                        info = _lib.RvSub()
                        dll.rv_sub_info(h, s, ctypes.byref(info))
                        idx._n_sub, idx._nsamples_sub = info.n, info.nsamples
                        if len(idx.skipmums) == 0:                                # reveal.c:802-837
                            mums, pre = idx._fetch_sub_mums(s, info), False
                        else:
                            mums, pre = idx.skipmums, True
                        res = mumpicker(mums, idx, precomputed=pre, minlength=int(minl))
                        if not isinstance(res, tuple):
                            raise TypeError("**** call to mumpicker failed")      # reveal.c:859-868
                        if len(res) == 0:
                            continue
                        mum, skipleft, skipright = res
                        l, mn, spd = mum                                          # reveal.c:901-925
                        sp = [int(spd[i][1]) for i in range(int(mn))]
                        r = align(idx, mum)                                       # reveal.c:939
                        if r is None:
                            continue
                        if not isinstance(r, tuple):
                            raise TypeError("**** call to graphalign failed")     # reveal.c:976-985
                        leading, trailing, matching, rest, merged, newleft, newright = r
                        la, nl = _pairs(leading, sa64); ta, nt = _pairs(trailing, sa64)
                        ma, nm = _pairs(matching, sa64); ra, nr = _pairs(rest, sa64)
                        spa = np.ascontiguousarray(np.array(sp, dtype=np.int64))
                        if dll.rv_sub_split(h, s, int(l), len(sp), spa.ctypes.data, la.ctypes.data, nl, ta.ctypes.data, nt,
                                            ma.ctypes.data, nm, ra.ctypes.data, nr) != 0:
                            self._fail()
                        decided[s] = (leading, trailing, rest, newleft, newright, skipleft, skipright)
                    nf = len(frontier)
                    kids = np.full(3 * max(nf, 1), -1, dtype=np.int32)
                    if dll.rv_frontier_commit(h, kids.ctypes.data) != 0:
                        self._fail()
                    nxt = {}
                    for s, (leading, trailing, rest, newleft, newright, skipleft, skipright) in decided.items():
                        p = frontier[s]
                        for kind, slot in enumerate(kids[3 * s:3 * s + 3]):
                            if slot < 0:
                                continue
                            c = index.__new__(index)                              # newIndex(), reveal.c:1136-1207
                            c._lib, c._dll, c._h, c._main = self._lib, dll, h, self
                            c._samples, c._constructed = self._samples, True
                            c._depth = p._depth + 1
                            c._slot, c._sx = int(slot), None
                            c._n_sub = c._nsamples_sub = None
                            if kind == 0:
                                c._nodes, c._leftnode, c._rightnode, c.skipmums = leading, p._leftnode, newright, skipleft
                            elif kind == 1:
                                c._nodes, c._leftnode, c._rightnode, c.skipmums = trailing, newleft, p._rightnode, skipright
                            else:
                                c._nodes, c._leftnode, c._rightnode, c.skipmums = rest, p._leftnode, p._rightnode, []
                            nxt[int(slot)] = c
                    frontier = [nxt[k] for k in sorted(nxt)]
                    for k, c in enumerate(frontier):
                        assert c._slot == k
                        if len(c.skipmums) != 0 and dll.rv_sub_skip_scan(h, k) != 0:      # seeded: not scanned (reveal.c:802, 830-837)
                            self._fail()
            finally:
                dll.rv_align_end(h)
                self._slot = None
                self._n_sub = self._nsamples_sub = None
            return None

        # ---- host-driven single steps (reveal.c:1386-1748, interface.c:432-470) ----------
        def _need_sx(self):
            if not self._constructed:
                raise TypeError("Index not yet constructed.")
            if self._slot is not None:
                raise error("not available on a sub-index handed to an align() callback")
            if not self._sx:
                self._sx = self._dll.rv_sx_main(self._h)
                if not self._sx:
                    self._fail()
            return self._sx

        def splitindex(self, leading_intervals, trailing_intervals, matching_intervals, rest, merged, newleftnode, newrightnode,
                       skipmumsleft, skipmumsright):
            """reveal.c:1515-1748 -> (leading, trailing, parallel) index objects or None.
            The Python-driven form of one recursion step (rem.py:580-609): this index keeps its SA / LCP."""
            sx = self._need_sx()
            la, nl = _pairs(leading_intervals, sa64); ta, nt = _pairs(trailing_intervals, sa64)
            ma, nm = _pairs(matching_intervals, sa64); ra, nr = _pairs(rest, sa64)
            out = (ctypes.c_void_p * 3)()
            if self._dll.rv_sx_split(sx, la.ctypes.data, nl, ta.ctypes.data, nt, ma.ctypes.data, nm, ra.ctypes.data, nr, out) != 0:
                self._fail()
            kids = []
            for kind in range(3):
                if not out[kind]:
                    kids.append(None)
                    continue
                c = index.__new__(index)                                          # newIndex(), reveal.c:1679-1735
                c._lib, c._dll, c._h, c._main = self._lib, self._dll, self._h, self._main
                c._samples, c._constructed = self._main._samples, True
                c._depth = self._depth + 1
                c._slot, c._sx = None, out[kind]
                c._n_sub = c._nsamples_sub = None
                if kind == 0:
                    c._nodes, c._leftnode, c._rightnode, c.skipmums = leading_intervals, self._leftnode, newrightnode, skipmumsleft
                elif kind == 1:
                    c._nodes, c._leftnode, c._rightnode, c.skipmums = trailing_intervals, newleftnode, self._rightnode, skipmumsright
                else:
                    c._nodes, c._leftnode, c._rightnode, c.skipmums = rest, self._leftnode, self._rightnode, []
                kids.append(c)
            return tuple(kids)

        def extract(self, intervals):
            """reveal.c:1386-1505: the suffixes of `intervals` leave this index (in place), the intervals are lower-cased
            in T.  After construct(rc=1) query-side intervals are mapped back and replaced in the list, as in the reference."""
            sx = self._need_sx()
            iv = list(intervals)
            a, k = _pairs(iv, sa64)
            before = a.copy()
            if self._dll.rv_sx_extract(sx, a.ctypes.data, k) != 0:
                self._fail()
            if isinstance(intervals, list):                                        # PyList_SetItem, reveal.c:1426
                for q in range(k):
                    if (a[q] != before[q]).any():
                        intervals[q] = (int(a[q][0]), int(a[q][1]))
            return None

        def copy(self):
            """interface.c:432-470: an independent copy (own text and arrays).  A sub-index copies its SA / LCP and
            goes on sharing the text of its main index."""
            if not self._constructed:
                raise TypeError("Index not yet constructed.")
            if self._slot is not None:
                raise error("not available on a sub-index handed to an align() callback")
            c = index.__new__(index)
            c._lib, c._dll = self._lib, self._dll
            c._safile, c._lcpfile, c._cache = "", "", 0
            c._samples, c._nodes, c.skipmums = list(self._samples), set(self._nodes), list(self.skipmums)
            c._leftnode, c._rightnode, c._depth = self._leftnode, self._rightnode, self._depth
            c._constructed, c._slot, c._sx = True, None, None
            c._n_sub = c._nsamples_sub = None
            c._rc = getattr(self, "_rc", 0)
            if self._main is self:
                if self._sx:
                    raise error("copy() after extract() / splitindex() on the main index is not supported")
                c._h = self._dll.rv_clone(self._h)
                if not c._h:
                    self._fail()
                c._main = c
            else:
                c._h, c._main = self._h, self._main
                c._samples = self._main._samples
                c._sx = self._dll.rv_sx_copy(self._sx)
                if not c._sx:
                    self._fail()
            return c

        def preselect(self, maxmums):
            """Not in the reference: from the next align() on, mumpicker
            receives of every scan only what schemes.graphmumpicker keeps of it before
            chaining -- the matches present in every sample of the sub-index
            (schemes.py:227), of those the `maxmums` longest (schemes.py:240, 287-289; of
            equal lengths the later emitted) -- still in emission order; a sub-index
            without such a match gets its whole list (schemes.py:229-232).  The selection
            happens inside the library, so the tuples of the others are never built.
            0 / None switches it off."""
            if self._depth != 0:
                raise error("preselect() is set on the main index")
            if self._dll.rv_set_preselect(self._h, int(maxmums or 0)) != 0:
                self._fail()

        def align_builtin(self, minl=20, minn=2, trace=False):
            """Not in the reference: the whole recursion with the library's
            built-in deterministic callbacks (longest full match, linear interval
            model) -- what bench.py times.  Same result as
            align(rem.bench_mumpicker, rem.linear_graphalign, minl=, minn=).
            -> dict(stats, anchors=[(l, (pos...))], trace=structured array | None)"""
            if not self._constructed:
                raise error("Index not yet constructed, alignment stopped.")
            dll, h = self._dll, self._h
            dll.rv_set_trace(h, 1 if trace else 0)
            self._offer_result_buffers()
            st = _lib.RvAlignStats()
            if dll.rv_align_builtin(h, int(minl), int(minn), ctypes.byref(st)) != 0:
                self._fail()
            return self._builtin_result(st, trace)

        def set_picker(self, args=None):
            """the picker of align_builtin (rv_set_picker): None = the benchmark picker; a schemes.PickerArgs = the reference's default picker
            (schemes.graphmumpicker) in C++ behind the ABI, for inputs with one sequence per sample (args.maxsize / maxdepth are not supported there)"""
            from . import schemes
            if args is None:
                r = self._dll.rv_set_picker(self._h, 0, None)
            else:
                if args.maxsize is not None or args.maxdepth is not None:
                    raise error("the native picker does not take --maxbubblesize / maxdepth")
                A = schemes._RvPickerArgs(int(args.wscore), int(args.wpen), int(args.maxmums or 0), int(args.seedsize or 0), schemes.GCMODELS[args.gcmodel],
                                          1 if args.trim else 0, float(args.pcutoff))
                r = self._dll.rv_set_picker(self._h, 1, ctypes.byref(A))
            if r != 0:
                self._fail()

        def set_replay_graph(self, graph=None):
            """rv_set_replay_graph: `graph` = alngraph.NativeGraph(G, root_nodes) without anchors; the next align_builtin with the native picker applies its anchors
            to it as the levels go by (a host thread beside the GPU's work); None = off"""
            if graph is not None and graph._dll is not self._dll:
                raise error("the graph was made by the other build of the library (32 / 64-bit suffix arrays)")
            if self._dll.rv_set_replay_graph(self._h, None if graph is None else graph._g) != 0:
                self._fail()

        def _sync_nodes(self):
            """sequences added inside the library (rv_graph_read_gfa) into this object's interval set"""
            self._nodes_stale = False
            k = self._dll.rv_nnodes(self._h)
            if k != len(self._nodes):
                buf = np.zeros(2 * max(k, 1), np.int64)
                if self._dll.rv_node_list(self._h, buf.ctypes.data) != 0:
                    self._fail()
                self._nodes = set(map(tuple, buf[:2 * k].reshape(-1, 2).tolist()))
                self._constructed = False

        def set_graph_picker(self, graph=None, args=None):
            """graph inputs (rv_set_graph_picker): `graph` = an alngraph.LoopGraph of the inputs made with this index' library; picker and graphalign of
            `reveal rem` run inside align_builtin on it, and it is the alignment graph afterwards.  None = off."""
            from . import schemes
            if graph is None:
                r = self._dll.rv_set_graph_picker(self._h, None, None)
            else:
                if args.maxsize is not None or args.maxdepth is not None:
                    raise error("the native picker does not take --maxbubblesize / maxdepth")
                if graph._dll is not self._dll:
                    raise error("the graph was made by the other build of the library (32 / 64-bit suffix arrays)")
                A = schemes._RvPickerArgs(int(args.wscore), int(args.wpen), int(args.maxmums or 0), int(args.seedsize or 0), schemes.GCMODELS[args.gcmodel],
                                          1 if args.trim else 0, float(args.pcutoff))
                r = self._dll.rv_set_graph_picker(self._h, graph._g, ctypes.byref(A))
            if r != 0:
                self._fail()

        def picker_info(self):
            o = (ctypes.c_int64 * 6)()
            self._dll.rv_picker_info(self._h, o)
            return dict(kind=int(o[0]), calls=int(o[1]), seeded=int(o[2]), picker_s=o[3] / 1e9, lists_s=o[4] / 1e9, graphalign_s=o[5] / 1e9)

        def _result_buffers_free(self):
            """the result arrays of the previous call, when nothing but this object refers to them (or to a view of them) any more"""
            c = self.__dict__.get("_res_bufs")
            if c is not None and sys.getrefcount(c[0]) == 2 and sys.getrefcount(c[1]) == 2 and sys.getrefcount(c[2]) == 2:
                return c
            return None

        def _offer_result_buffers(self):
            """rv_set_result_buffers: the run delivers its anchors straight into the arrays the caller has let go of (page-locked by the
            library while they are set); arrays somebody still holds are taken back from the library first"""
            c = self._result_buffers_free()
            if c is not None:
                self._dll.rv_set_result_buffers(self._h, c[0].ctypes.data, len(c[0]), c[1].ctypes.data, len(c[1]), c[2].ctypes.data, len(c[2]))
            else:
                self._dll.rv_set_result_buffers(self._h, None, 0, None, 0, None, 0)

        def _builtin_result(self, st, trace):
            dll, h = self._dll, self._h
            mem = ctypes.c_int64(0)
            na = dll.rv_anchor_count(h, ctypes.byref(mem))
            # (filled completely by the library: no zeroing.)  The arrays of the previous call are used again when the caller has let go of
            # them -- nothing else refers to them or to a view of them: 56 MB of fresh pages per call cost 2 x 250 Mbp 1-3 ms of page faults
            c = self._result_buffers_free()
            if c is not None and len(c[0]) >= max(na, 1) and len(c[1]) >= na + 1 and len(c[2]) >= max(mem.value, 1):
                l, off, pos = c
            else:
                dll.rv_set_result_buffers(h, None, 0, None, 0, None, 0)      # (the arrays that are replaced may be freed: not the library's any more)
                if os.environ.get("RV_RESULT_BUFS", "") == "heap":      # (diagnostics: arrays from the C heap are never page-locked by the library)
                    l = np.empty(max(na, 1), dtype=np.uint32); off = np.empty(na + 1, dtype=np.int64); pos = np.empty(max(mem.value, 1), dtype=np.int64)
                else:
                    l = _page_array(max(na, 1), np.uint32); off = _page_array(na + 1, np.int64); pos = _page_array(max(mem.value, 1), np.int64)
                self.__dict__["_res_bufs"] = (l, off, pos)
            c = None
            off[0] = 0
            if dll.rv_fetch_anchors(h, l.ctypes.data, off.ctypes.data, pos.ctypes.data) != 0:
                self._fail()
            tr = None
            if trace:
                nt = dll.rv_trace_count(h)
                tr = np.zeros(max(nt, 1), dtype=_lib.TRACE_DTYPE)
                if dll.rv_fetch_trace(h, tr.ctypes.data, len(tr)) != 0:
                    self._fail()
                tr = tr[:nt]
            stats = {f[0]: getattr(st, f[0]) for f in _lib.RvAlignStats._fields_}
            return dict(stats=stats, anchors=(l[:na], off[:na + 1], pos[:mem.value]), trace=tr)

        # ---- one alignment over several devices (include/reveal_amd.h "frontier hand-off"; reveal_amd/shard.py drives it)
        @property
        def maxlcp(self):
            return int(self._dll.rv_maxlcp(self._h))

        def cascade_info(self):
            """what the anchor cascade did in the last align_builtin (rv_cascade_info)"""
            o = np.zeros(8, dtype=np.int64)
            self._dll.rv_cascade_info(self._h, o.ctypes.data)
            return dict(done=bool(o[0]), levels=int(o[1]), matches=int(o[2]), witnesses=int(o[3]), subindices=int(o[4]), undecided=int(o[5]), rebuilt_ranks=int(o[6]), decided_from_witnesses=int(o[7]),
                        why=(self._dll.rv_cascade_why(self._h) or b"").decode())

        def align_builtin_until(self, stop_subs, minl=20, minn=2, trace=False):
            """align_builtin that stops once the frontier holds >= stop_subs sub-indices.
            -> frontier size (0: the run finished first; align_builtin_resume() still returns its result)"""
            if not self._constructed:
                raise error("Index not yet constructed, alignment stopped.")
            self._dll.rv_set_trace(self._h, 1 if trace else 0)
            self._offer_result_buffers()      # (arrays of an earlier result the caller still holds are taken back from the library: the run writes into what is set)
            self._st = _lib.RvAlignStats()
            self._trace = bool(trace)
            r = self._dll.rv_align_builtin_until(self._h, int(minl), int(minn), int(stop_subs), ctypes.byref(self._st))
            if r < 0:
                self._fail()
            self._pending = r > 0
            return r

        def align_builtin_continue(self, stop_subs):
            """widen the frontier of a run stopped by align_builtin_until: at least one more level, until it holds >= stop_subs
            sub-indices.  -> new frontier size (0: finished on the way)"""
            if not getattr(self, "_pending", False):
                return 0
            self._offer_result_buffers()
            r = self._dll.rv_align_builtin_continue(self._h, int(stop_subs), ctypes.byref(self._st))
            if r < 0:
                self._fail()
            self._pending = r > 0
            return r

        def frontier(self):
            """-> dict(level, m, meta[nsubs,6] = (offset, n, depth, nsamples, kind, parent), node_first[nsubs+1], nodes[nnodes,2])"""
            c = np.zeros(4, dtype=np.int64)
            if self._dll.rv_frontier_counts(self._h, c.ctypes.data) != 0:
                self._fail()
            ns, m, nn, level = (int(x) for x in c)
            meta = np.zeros((max(ns, 1), 6), dtype=np.int64); nf = np.zeros(ns + 1, dtype=np.int64); nodes = np.zeros((max(nn, 1), 2), dtype=np.int64)
            if self._dll.rv_frontier_export(self._h, meta.ctypes.data, nf.ctypes.data, nodes.ctypes.data) != 0:
                self._fail()
            return dict(level=level, m=m, meta=meta[:ns], node_first=nf, nodes=nodes[:nn])

        def frontier_pack(self, subs, sa, lcp, bwt):
            """segments of the sub-indices `subs` back to back into sa / lcp / bwt: numpy arrays (host) or torch
            tensors of this device (dtype = the library's SA / LCP types, uint8); -> ranks written"""
            subs = np.ascontiguousarray(subs, dtype=np.int32)
            ptr, dev = _pointers(sa, lcp, bwt)
            r = self._dll.rv_frontier_pack(self._h, subs.ctypes.data, len(subs), ptr[0], ptr[1], ptr[2], dev)
            if r < 0:
                self._fail()
            return int(r)

        def frontier_seeds(self, subs):
            """rv_set_picker(1) runs: the seed lists the native picker left for the sub-indices `subs` of the frontier (the reference's
            skipmums, reveal.c:1157, 1180), packed as int64 words for frontier_import(part with part["seeds"] = these)"""
            subs = np.ascontiguousarray(subs, dtype=np.int32)
            need = self._dll.rv_frontier_seeds_export(self._h, subs.ctypes.data, len(subs), None, 0)
            if need < 0:
                self._fail()
            out = np.zeros(max(int(need), 1), dtype=np.int64)
            if self._dll.rv_frontier_seeds_export(self._h, subs.ctypes.data, len(subs), out.ctypes.data, int(need)) != need:
                self._fail()
            return out[:int(need)]

        def frontier_import(self, part, sa, lcp, bwt, minl=20, minn=2, maxlcp=None, trace=False):
            """make `part` (a subset of a frontier() dict: level, meta, node_first, nodes; optionally "seeds" = frontier_seeds() of the
            same sub-indices) with its packed segments the frontier of this index.  An index that only holds its samples (no construct)
            becomes a worker."""
            meta = np.ascontiguousarray(part["meta"], dtype=np.int64).reshape(-1, 6)
            nf = np.ascontiguousarray(part["node_first"], dtype=np.int64); nodes = np.ascontiguousarray(part["nodes"], dtype=np.int64).reshape(-1, 2)
            m = int(meta[:, 1].sum()) if len(meta) else 0
            ptr, dev = _pointers(sa, lcp, bwt)
            self._offer_result_buffers()
            if not getattr(self, "_pending", False):
                self._dll.rv_set_trace(self._h, 1 if trace else 0)
                self._st = _lib.RvAlignStats()
                self._trace = bool(trace)
            ml = self.maxlcp if maxlcp is None else int(maxlcp)
            if self._dll.rv_frontier_import(self._h, int(minl), int(minn), ml, int(part.get("level", 1)), len(meta), meta.ctypes.data, nf.ctypes.data,
                                            nodes.ctypes.data, m, ptr[0], ptr[1], ptr[2], dev) != 0:
                self._fail()
            self._pending = True
            seeds = part.get("seeds")
            if seeds is not None and len(seeds) > len(meta):      # (more than the zero per sub-index of "no seeds")
                seeds = np.ascontiguousarray(seeds, dtype=np.int64)
                if self._dll.rv_frontier_seeds_import(self._h, len(meta), seeds.ctypes.data, len(seeds)) != 0:
                    self._fail()

        def align_builtin_resume(self):
            """finish the run started by align_builtin_until / frontier_import; result as align_builtin"""
            if getattr(self, "_pending", False):
                self._offer_result_buffers()
                if self._dll.rv_align_builtin_resume(self._h, ctypes.byref(self._st)) != 0:
                    self._fail()
            self._pending = False
            return self._builtin_result(self._st, self._trace)

        def _fetch_sub_mums(self, s, info):
            cnt, mem = info.nmums, info.nmembers
            l = np.zeros(max(cnt, 1), dtype=np.uint32); n = np.zeros(max(cnt, 1), dtype=np.int32)
            off = np.zeros(cnt + 1, dtype=np.int64)
            so = np.zeros(max(mem, 1), dtype=np.uint16); pos = np.zeros(max(mem, 1), dtype=np.int64)
            if self._dll.rv_sub_mums(self._h, s, l.ctypes.data, n.ctypes.data, off.ctypes.data, so.ctypes.data, pos.ctypes.data) != 0:
                self._fail()
            return _csr_to_tuples(cnt, l, n, off, so, pos)

        def __reduce__(self):                                # interface.c:417-422 (stub there too)
            return None

    index.__name__ = "index"
    index.__qualname__ = "index"
    return index


def _pointers(*bufs):
    """addresses of numpy arrays or torch tensors, and whether they are device memory"""
    ptr, dev = [], None
    for b in bufs:
        on_dev = bool(getattr(b, "is_cuda", False))
        if dev is not None and on_dev != dev:
            raise ValueError("sa, lcp and bwt must live in the same kind of memory")
        dev = on_dev
        if hasattr(b, "data_ptr"):
            if not b.is_contiguous():
                raise ValueError("contiguous buffers needed")
            ptr.append(b.data_ptr())
        else:
            if not b.flags["C_CONTIGUOUS"]:
                raise ValueError("contiguous buffers needed")
            ptr.append(b.ctypes.data)
    return ptr, 1 if dev else 0


def _csr_to_tuples(cnt, l, n, off, so, pos):
    l, n, off = l.tolist(), n.tolist(), off.tolist()
    members = list(zip(so.tolist(), pos.tolist()))      # (one pass in C for the pairs, a slice per match: 2.6 x the speed of a generator per match -- 10^7 members took 12 s)
    return [(l[k], n[k], tuple(members[off[k]:off[k + 1]])) for k in range(cnt)]
