"""CPU: host logic of com_amd.train (no GPU, no library calls): the loop of train_one_epoch (tools/train_utils/train_utils.py:60-95)
over a step object, the OneCycle scheduler against the reference fixture G8, the scoped static plan, the execution switches."""
import gc

import numpy as np
import pytest


class _FakeStep:
    def __init__(self, overflow_at=None):
        self.calls, self.lr_steps, self.polls = [], [], 0
        self.overflow_at = overflow_at
        self.lr_scheduler = self

    def step(self, accumulated_iter=None):                  # (the lr_scheduler role)
        self.lr_steps.append(accumulated_iter)

    def __call__(self, batch, staged=None):
        assert not gc.isenabled(), "the loop must keep the cyclic collector off while it issues steps"
        self.calls.append(batch)
        if staged is not None:
            staged()

    def poll(self):
        self.polls += 1
        return self.overflow_at is not None and len(self.calls) >= self.overflow_at


def test_train_one_epoch_loop_shape_and_overflow():
    from com_amd import train
    from com_amd import _lib as L
    st = _FakeStep()
    staged = []
    n = train.train_one_epoch(st, ["b0", "b1", "b2"], 7, accumulated_iter=10, on_staged=lambda: staged.append(1))
    assert n == 17 and st.lr_steps == list(range(10, 17))
    assert st.calls == ["b1", "b2", "b0", "b1", "b2", "b0", "b1"]        # the data side runs one batch ahead of prime()'s batch 0
    assert len(staged) == 7 and st.polls == 0 and gc.isenabled()
    it = iter(range(100))
    st2 = _FakeStep()
    assert train.train_one_epoch(st2, it, 16) == 16 and st2.calls == list(range(16)) and st2.polls == 2
    with pytest.raises(L.PcdError):
        train.train_one_epoch(_FakeStep(overflow_at=8), ["a", "b"], 32)
    assert gc.isenabled()                                                  # restored on the error path too


def test_one_cycle_scheduler_steps_like_the_reference(golden):
    from com_amd import train

    class Opt:
        def __init__(self):
            self.hyper, self.table = [], None

        def set_hyper(self, lr, mom):
            self.hyper.append((lr, mom))

        def set_schedule(self, pairs):
            self.table = np.asarray(pairs)

    g = golden("g8_adam_onecycle")
    total = int(g["total_steps"][0])
    o = Opt()
    sch = train.OneCycle(o, total, device_table=False)
    for it in range(total):
        sch.step(it)
    np.testing.assert_allclose(np.array(o.hyper)[:, 0], g["lr"], rtol=1e-12)
    np.testing.assert_allclose(np.array(o.hyper)[:, 1], g["mom"], rtol=1e-12)
    o2 = Opt()
    sch2 = train.OneCycle(o2, total)                        # the device-table form moves nothing per step
    sch2.step()
    sch2.step()
    assert sch2.last_iter == 1 and not o2.hyper
    np.testing.assert_allclose(o2.table[:, 0], g["lr"], rtol=1e-12)


def test_static_plan_is_scoped_not_global():
    from com_amd import ops
    a, b = ops.StaticPlan(), ops.StaticPlan()
    assert ops.current_plan() is None
    with a:
        assert ops.current_plan() is a
        with b:
            assert ops.current_plan() is b
        assert ops.current_plan() is a
    assert ops.current_plan() is None and not hasattr(ops, "PLAN")
    a.observe("voxels", 1000)
    assert a.cap("voxels") == 2048 and not b.caps            # 1.25 x, rounded up to 1024; another plan sees nothing
    a.grow(1.5)
    assert a.cap("voxels") == 2048 and a.caps["voxels"] == 1501
    with pytest.raises(ops.L.PcdError):
        b.cap("voxels")


def test_execution_switches_are_restored():
    from com_amd import train
    from com_amd.spconv import functional as Fsp
    before = (Fsp.DIRECT_GRAD, Fsp.WGRAD_JOIN_LAG, Fsp.FUSE_BN_REDUCTIONS)
    opts = train._ExecOptions({"WGRAD_JOIN_LAG": 5})
    with opts:
        assert Fsp.DIRECT_GRAD is True and Fsp.WGRAD_JOIN_LAG == 5
        with pytest.raises(RuntimeError):
            with train._ExecOptions():
                assert Fsp.WGRAD_JOIN_LAG == 32
                raise RuntimeError("inside")
        assert Fsp.WGRAD_JOIN_LAG == 5
    assert (Fsp.DIRECT_GRAD, Fsp.WGRAD_JOIN_LAG, Fsp.FUSE_BN_REDUCTIONS) == before
