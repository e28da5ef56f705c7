"""Point-to-point transport of the frontier hand-off without a tensor library ("no RCCL; only point-to-point
copies of child arrays; results gathered on host").

`Group` is the ranks of one node talking over local sockets (multiprocessing.connection: rank 0 listens, the others connect):
requests, metadata and results are pickled messages; rank 0 waits on all its connections at once (no store to poll).  The
segments themselves -- SA / LCP / BWT of the sub-indices a worker takes, 9 B per rank -- do not go through the sockets when
the ranks own GPUs: rank 0 packs them into buffers of its device once, exports those (`DeviceMemory.export`: hipIpcGetMemHandle,
64 bytes per buffer), and a worker opens them on its own device and copies its batches out, device to device over the peer
link (include/reveal_amd.h rv_dev_* / rv_ipc_*).  `HostMemory` is the same interface over numpy arrays that travel inside the
messages (ranks without a GPU each: the protocol tests), `SharedMemory` the same over POSIX shared memory -- the device branch of
reveal_amd/shard.py executed without a device (tests/test_cpu_host.py).

torch.distributed is not needed for any of this; bench.py keeps it for the barrier / max-over-ranks timing its contract names.
"""
import os
import time
from multiprocessing.connection import Client, Listener, wait

import numpy as np

_AUTH = b"reveal_amd frontier hand-off"


class Group:
    """ranks 0 .. world-1 of one node.  rank 0: listen(); the others: connect()."""

    def __init__(self, rank, world, addr="127.0.0.1", port=None, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        if port is None:
            port = int(os.environ.get("RV_SHARD_PORT", 0)) or int(os.environ.get("MASTER_PORT", "29500")) + 101
        self.address = (addr, int(port))
        self.conns = {}            # rank 0: worker rank -> connection; workers: {0: connection}
        self._listener = None
        if self.world <= 1:
            return
        if self.rank == 0:
            self._listener = Listener(self.address, authkey=_AUTH)
            while len(self.conns) < self.world - 1:
                c = self._listener.accept()
                self.conns[int(c.recv())] = c
        else:
            t0 = time.time()
            while True:
                try:
                    c = Client(self.address, authkey=_AUTH)
                    break
                except (ConnectionRefusedError, FileNotFoundError, OSError):
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.05)
            c.send(self.rank)
            self.conns[0] = c

    @classmethod
    def from_env(cls, **kw):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), os.environ.get("MASTER_ADDR", "127.0.0.1"), **kw)

    # ---- rank 0
    def ready(self, timeout=None):
        """-> [(worker, message)] of every connection that has something to say (blocks up to `timeout` seconds; None: until one has)"""
        out = []
        by_conn = {c: w for w, c in self.conns.items()}
        for c in wait(list(by_conn), timeout):
            out.append((by_conn[c], c.recv()))
        return out

    def send(self, to, obj):
        self.conns[to].send(obj)

    # ---- workers
    def ask(self, obj):
        self.conns[0].send(obj)
        return self.conns[0].recv()

    def recv(self, frm=0):
        return self.conns[frm].recv()

    def barrier(self):
        if self.world <= 1:
            return
        if self.rank == 0:
            for w in self.conns:
                assert self.conns[w].recv() == "barrier"
            for w in self.conns:
                self.conns[w].send("go")
        else:
            assert self.ask("barrier") == "go"

    def close(self):
        for c in self.conns.values():
            c.close()
        self.conns = {}
        if self._listener is not None:
            self._listener.close()
            self._listener = None


class _View:
    """a stretch of a transport buffer: what index.frontier_pack / frontier_import take (duck-typed like a tensor: data_ptr, is_cuda, is_contiguous)"""

    def __init__(self, base, ptr, n, itemsize, on_device, keep=None):
        self.base, self.ptr, self.n, self.itemsize, self.is_cuda, self._keep = base, ptr, n, itemsize, on_device, keep

    def data_ptr(self):
        return self.ptr

    def is_contiguous(self):
        return True

    def __len__(self):
        return self.n

    def __getitem__(self, sl):
        lo, hi, step = sl.indices(self.n)
        assert step == 1
        return _View(self.base, self.ptr + lo * self.itemsize, max(hi - lo, 0), self.itemsize, self.is_cuda, self._keep)

    @property
    def nbytes(self):
        return self.n * self.itemsize


class DeviceMemory:
    """buffers of this process' GPU, shared with the other ranks through HIP's inter-process handles"""
    kind = "device"

    def __init__(self, lib, device):
        self.lib, self.dll, self.device = lib, lib.dll, int(device)
        self._mine, self._opened = [], []

    def alloc(self, n, itemsize):
        p = self.dll.rv_dev_alloc(self.device, int(max(n, 1)) * itemsize)
        if not p:
            raise MemoryError(self.lib.err())
        self._mine.append(p)
        return _View(p, p, n, itemsize, True)

    def export(self, view):
        import ctypes
        h = (ctypes.c_uint8 * 64)()
        if self.dll.rv_ipc_export(self.device, view.base, h) != 0:
            raise RuntimeError(self.lib.err())
        return ("ipc", bytes(h), view.ptr - view.base, view.n, view.itemsize)

    def open(self, token):
        import ctypes
        _, h, off, n, itemsize = token
        buf = (ctypes.c_uint8 * 64).from_buffer_copy(h)
        p = self.dll.rv_ipc_open(self.device, buf)
        if not p:
            raise RuntimeError(self.lib.err())
        self._opened.append(p)
        return _View(p, p + off, n, itemsize, True)

    def copy(self, dst, src, n):
        if n and self.dll.rv_dev_copy(self.device, dst.ptr, src.ptr, int(n) * dst.itemsize) != 0:
            raise RuntimeError(self.lib.err())

    def release(self):
        for p in self._opened:
            self.dll.rv_ipc_close(self.device, p)
        for p in self._mine:
            self.dll.rv_dev_free(self.device, p)
        self._mine, self._opened = [], []


class SharedMemory:
    """the device branch without a device: POSIX shared memory stands in for HBM, its name for the IPC handle (protocol tests)"""
    kind = "device"

    def __init__(self):
        self._segs = []

    def alloc(self, n, itemsize):
        from multiprocessing import shared_memory
        seg = shared_memory.SharedMemory(create=True, size=max(int(n), 1) * itemsize)
        self._segs.append((seg, True))
        arr = np.frombuffer(seg.buf, dtype=np.uint8)
        return _View(arr.ctypes.data, arr.ctypes.data, n, itemsize, False, keep=(seg, arr))

    def export(self, view):
        seg = view._keep[0]
        return ("shm", seg.name, view.ptr - view.base, view.n, view.itemsize)

    def open(self, token):
        from multiprocessing import shared_memory
        _, name, off, n, itemsize = token
        seg = shared_memory.SharedMemory(name=name)
        self._segs.append((seg, False))
        arr = np.frombuffer(seg.buf, dtype=np.uint8)
        return _View(arr.ctypes.data, arr.ctypes.data + off, n, itemsize, False, keep=(seg, arr))

    def copy(self, dst, src, n):
        import ctypes
        ctypes.memmove(dst.ptr, src.ptr, int(n) * dst.itemsize)

    def release(self):
        for seg, mine in self._segs:
            try:
                seg.close()
                if mine:
                    seg.unlink()
            except Exception:      # noqa: BLE001  (views of the buffer may still be alive: the segment goes with the process)
                pass
        self._segs = []


class HostMemory:
    """numpy arrays; a batch's segments travel inside the reply (ranks that share no memory of any kind)"""
    kind = "host"

    def alloc(self, n, itemsize):
        arr = np.zeros(max(int(n), 1) * itemsize, dtype=np.uint8)
        return _View(arr.ctypes.data, arr.ctypes.data, n, itemsize, False, keep=(None, arr))

    def bytes_of(self, view):
        import ctypes
        return ctypes.string_at(view.ptr, view.n * view.itemsize)

    def fill(self, view, data):
        import ctypes
        ctypes.memmove(view.ptr, data, len(data))

    def release(self):
        pass
