"""The MAPs context pool's bookkeeping (hashgan_amd.metric._Pool) with a stand-in engine: no GPU, no library -- who gets which context,
what goes back, what is destroyed.  (The same with real contexts: tests/test_hip_parity.py::test_a_new_maps_object_per_evaluation_recycles_one_context.)"""
import threading

import pytest

from hashgan_amd import metric


class _Ctx:
    def __init__(self, device):
        self.device = device
        self.options_touched = set()
        self._h = object()
        self.closed = False
        self.preloaded = False

    def preload(self):
        self.preloaded = True

    def get_stat(self, key):
        return 1000 if key == "device_bytes" else 0

    def close(self):
        self.closed = True
        self._h = None


class _Engine:
    made = []

    def __init__(self, device=0):
        self.ctx = _Ctx(device)
        self.b = self.C = self.N = self.db_kind = self.db_src = None
        self.resident = None
        _Engine.made.append(self)

    def forget(self):
        self.b = self.C = self.N = self.db_kind = self.db_src = None
        self.resident = None

    def close(self):
        self.ctx.close()


@pytest.fixture()
def pool(monkeypatch):
    monkeypatch.setattr(metric, "RetrievalEngine", _Engine)
    P = metric._Pool
    saved = (dict(P.idle), P.created, P.recycled, P.closed)
    P.idle, P.created, P.recycled, P.closed = {}, 0, 0, 0
    _Engine.made = []
    yield P
    P.idle, P.created, P.recycled, P.closed = saved


def test_an_object_borrows_on_first_use_and_hands_back_on_close(pool):
    m = metric.MAPs(10)
    assert pool.stats()["contexts_created"] == 0           # nothing until the first use
    e = m._engine()
    assert e.ctx.preloaded and pool.stats()["contexts_created"] == 1 and m._engine() is e
    e.b = 64
    m.close()
    st = pool.stats()
    assert st["contexts_idle"] == 1 and st["idle_device_bytes"] == 1000 and e.b is None and not e.ctx.closed
    m2 = metric.MAPs(5)
    assert m2._engine() is e and pool.stats()["contexts_recycled"] == 1 and pool.stats()["contexts_idle"] == 0
    m2.close()
    m2.close()                                             # closing twice is harmless
    assert pool.stats()["contexts_idle"] == 1


def test_live_objects_never_share_and_the_pool_keeps_at_most_two(pool):
    ms = [metric.MAPs(3) for _ in range(4)]
    es = [m._engine() for m in ms]
    assert len({id(e) for e in es}) == 4
    for m in ms:
        m.close()
    st = pool.stats()
    assert st["contexts_idle"] == pool.max_idle == 2 and st["contexts_closed"] == 2
    assert sum(e.ctx.closed for e in es) == 2
    metric._Pool.close_all()
    assert pool.stats()["contexts_idle"] == 0 and all(e.ctx.closed for e in es)


def test_changed_options_pending_steps_and_hip_errors_end_a_context(pool):
    for spoil in ("option", "flight", "poison", "keep_floats"):
        m = metric.MAPs(3)
        e = m._engine()
        if spoil == "option":
            e.ctx.options_touched.add("guess_sigma")
        elif spoil == "flight":
            e.ctx._in_flight = [64]
        elif spoil == "poison":
            e.poisoned = True
        else:
            e.ctx.options_touched.add("keep_floats")       # every load sets it: not the borrower's doing
        m.close()
        assert e.ctx.closed == (spoil != "keep_floats"), spoil
    assert pool.stats()["contexts_idle"] == 1


def test_devices_have_pools_of_their_own_and_threads_may_race(pool):
    a, b = metric.MAPs(1, device=0), metric.MAPs(1, device=1)
    ea, eb = a._engine(), b._engine()
    a.close(); b.close()
    assert metric.MAPs(1, device=1)._engine() is eb and metric.MAPs(1, device=0)._engine() is ea
    seen, lock = [], threading.Lock()

    def work():
        for _ in range(200):
            m = metric.MAPs(2)
            e = m._engine()
            with lock:
                assert id(e) not in seen
                seen.append(id(e))
            with lock:
                seen.remove(id(e))
            m.close()
    th = [threading.Thread(target=work) for _ in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    st = pool.stats()
    assert st["contexts_idle"] <= 2 * pool.max_idle and st["contexts_created"] - st["contexts_closed"] == st["contexts_idle"]
