"""A batch of independent alignments on one GPU (include/reveal_amd.h rv_batch_*; the reference's counterpart: the independent `reveal rem` commands
`reveal align` prints, reveal/align.py:27-54).

    res = batch.Batch(indices).run(minl=20, minn=2)        # [align_builtin-shaped result per index, in order]

Every index is constructed (unless construct=False) and aligned with the built-in callbacks on a host thread of its own inside the library; the jobs with
more than two samples run the level loops of their anchor cascades as one set of launches.  Results are those of `index.align_builtin` on each."""
import ctypes

import numpy as np

from . import _lib


class Batch:
    def __init__(self, indices):
        self.indices = list(indices)
        if not self.indices:
            raise ValueError("an empty batch")
        lib = self.indices[0]._lib
        if any(ix._lib is not lib for ix in self.indices):
            raise ValueError("the indices of a batch come from one module (reveallib or reveallib64)")
        self._lib, self._dll = lib, lib.dll
        self._b = self._dll.rv_batch_new()
        if not self._b:
            raise MemoryError(lib.err())
        for ix in self.indices:
            if self._dll.rv_batch_add(self._b, ix._h) != 0:
                raise RuntimeError(lib.err())

    def run(self, minl=20, minn=2, construct=True):
        n = len(self.indices)
        stats = (_lib.RvAlignStats * n)()
        status = (ctypes.c_int * n)()
        for ix in self.indices:
            if not construct and not ix._constructed:
                raise ix._error("Index not yet constructed, alignment stopped.") if hasattr(ix, "_error") else RuntimeError("Index not yet constructed")
            self._dll.rv_set_trace(ix._h, 0)
            ix._offer_result_buffers()
        r = self._dll.rv_batch_run(self._b, int(minl), int(minn), 1 if construct else 0, stats, status)
        if r != 0:
            raise RuntimeError(self._lib.err())
        out = []
        for i, ix in enumerate(self.indices):
            if construct:
                ix._constructed = True; ix._rc = 0; ix._depth = 0; ix._main = ix
            out.append(ix._builtin_result(stats[i], False))
        return out

    def info(self):
        o = np.zeros(2, dtype=np.int64)
        self._dll.rv_batch_info(self._b, o.ctypes.data)
        return dict(joint_level_loops=int(o[0]), jobs_served=int(o[1]))

    def close(self):
        if getattr(self, "_b", None):
            self._dll.rv_batch_free(self._b)
            self._b = None

    def __del__(self):
        self.close()
