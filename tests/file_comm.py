"""A communicator that goes through files -- TEST INFRASTRUCTURE for multi-PROCESS dry runs of the sharded sequence
and of bench.py's N > 1 leg on a box with ONE GPU (RCCL refuses two ranks on one device): every rank is its own
process with its own context on GPU 0, all_gather = device -> host -> file, a file barrier, files -> host -> device.
Slow by construction; it exists to exercise the control flow of a multi-rank run (rendezvous, shard bounds, the
orchestration's branches, the max-over-ranks clock, who prints), never to be measured."""
import os
import time

import numpy as np

from hashgan_amd.sharded import DevBuf


class FileComm:
    def __init__(self, directory, rank, world, ctx, timeout=600.0):
        self.dir, self.rank, self.world, self.ctx, self.timeout = directory, int(rank), int(world), ctx, timeout
        self._n = 0
        self._slot = 0
        os.makedirs(directory, exist_ok=True)

    def _path(self, tag, rank):
        return os.path.join(self.dir, "%s_%d_r%d" % (tag, self._n, rank))

    def _wait(self, path):
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > self.timeout:
                raise RuntimeError("rank %d: timed out waiting for %s" % (self.rank, path))
            time.sleep(0.002)

    def _publish(self, tag, payload):
        tmp = self._path(tag, self.rank) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, self._path(tag, self.rank))

    def _collect(self, tag):
        out = []
        for r in range(self.world):
            self._wait(self._path(tag, r))
            with open(self._path(tag, r), "rb") as f:
                out.append(f.read())
        return out

    def all_gather(self, buf):
        host = np.empty(buf.nbytes, np.uint8)
        self.ctx.memcpy_dtoh(host, buf.ptr, buf.nbytes)
        self._publish("ag", host.tobytes())
        parts = self._collect("ag")
        self._n += 1
        slot = self._slot
        self._slot = (slot + 1) % 4
        base = self.ctx.scratch(slot, buf.nbytes * self.world)
        allb = np.frombuffer(b"".join(parts), np.uint8)
        assert allb.size == buf.nbytes * self.world
        self.ctx.memcpy_htod(base, np.ascontiguousarray(allb), allb.size)
        return DevBuf(base, allb.size)

    def all_to_all(self, buf):
        host = np.empty(buf.nbytes, np.uint8)
        self.ctx.memcpy_dtoh(host, buf.ptr, buf.nbytes)
        self._publish("a2a", host.tobytes())
        parts = self._collect("a2a")
        self._n += 1
        n = buf.nbytes // self.world
        mine = np.frombuffer(b"".join(p[self.rank * n:(self.rank + 1) * n] for p in parts), np.uint8)
        slot = self._slot
        self._slot = (slot + 1) % 4
        base = self.ctx.scratch(slot, buf.nbytes)
        self.ctx.memcpy_htod(base, np.ascontiguousarray(mine), mine.size)
        return DevBuf(base, mine.size)

    def barrier(self):
        self._publish("bar", b"x")
        self._collect("bar")
        self._n += 1

    def allreduce_max(self, x):
        self._publish("max", np.float64(x).tobytes())
        vals = [np.frombuffer(p, np.float64)[0] for p in self._collect("max")]
        self._n += 1
        return float(max(vals))

    def all_gather_host(self, arr):
        arr = np.ascontiguousarray(arr)
        self._publish("agh", arr.tobytes())
        parts = self._collect("agh")
        self._n += 1
        return np.stack([np.frombuffer(p, arr.dtype).reshape(arr.shape) for p in parts])
