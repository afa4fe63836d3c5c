"""Per-network launch programs (rg_hip/netgraph.py): the dual_gan networks' forward / backward programs replayed as captured
single-stream hipGraphs give the SAME numbers as the eager launches, bit for bit, step after step; networks the capture cannot
represent (active Dropout) stay eager; a record in flight is never reused."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda", 0)


def _dptn_run(dev, graphs, steps, conv_dtype):
    import bench
    import rg_hip.netgraph as NG
    from dual_gan.models.models import create_model
    old = NG.ENABLED
    NG.ENABLED = graphs
    try:
        torch.manual_seed(4321)
        gan = create_model(bench.dualgan_opt(model="DPTN", model_gen="DPTN", gan_mode="hinge", layers_g=3, conv_dtype=conv_dtype,
                                             lambda_rec=5.0, lambda_g=2.0, t_s_ratio=0.5, dis_layers=4, ratio_g2d=0.1))
        losses = []
        for i in range(steps):
            gan.set_input(bench.synth_dualgan(4, dev, 100 + i, with_target=True))
            gan.optimize_parameters()
            losses.append({k: float(v) for k, v in gan.get_current_errors().items()})
        params = [p.detach().clone() for p in gan.net_G.parameters()] + [p.detach().clone() for p in gan.net_D.parameters()]
        fake = gan.fake_image_t.detach().clone() if hasattr(gan, "fake_image_t") else None
        st = {n: NG.stats(getattr(getattr(gan, n), "module", getattr(gan, n))) for n in ("net_G", "net_D")}
        return losses, params, fake, st
    finally:
        NG.ENABLED = old


# default run: the fp8 variant (same capture machinery + the fp8 caches); fp32 graphed == eager is also test_joint_4a_step_graphed_equal_eager
@pytest.mark.parametrize("conv_dtype", [pytest.param("fp32", marks=pytest.mark.slow), "fp8"])
def test_dptn_steps_graphed_equal_eager_bit_for_bit(dev, conv_dtype):
    steps = 6
    le, pe, fe, _ = _dptn_run(dev, False, steps, conv_dtype)
    lg, pg, fg, st = _dptn_run(dev, True, steps, conv_dtype)
    assert le == lg, [(i, a, b) for i, (a, b) in enumerate(zip(le, lg)) if a != b][:2]
    assert all(torch.equal(a, b) for a, b in zip(pe, pg))
    if fe is not None:
        assert torch.equal(fe, fg)
    # the programs really were captured and replayed: every key that was called more than WARMUP times owns records
    import rg_hip.netgraph as NG
    for name, s in st.items():
        assert s, name
        for key, (calls, records, ok) in s.items():
            assert ok, (name, key)
            if calls > NG.WARMUP:
                assert 1 <= records <= NG.MAX_RECORDS, (name, key, calls, records)


def test_joint_4a_step_graphed_equal_eager(dev):
    import bench
    import rg_hip.netgraph as NG

    def run(graphs):
        old = NG.ENABLED
        NG.ENABLED = graphs
        try:
            cls = bench.WORKLOADS["4a"]
            cls.crops = 8
            w = cls()
            w.build(dev, 0)
            out = []
            for _ in range(5):
                w.step()
                out.append({k: float(v) for k, v in w.losses().items()})
            return out
        finally:
            NG.ENABLED = old
    assert run(False) == run(True)


def test_dropout_network_is_not_graphed_and_busy_records_are_not_reused(dev):
    """(a) a network with active Dropout runs eagerly however often it is called; (b) two forwards of one graphed network before
    any backward use two records (the first one's saved activations must survive the second forward)."""
    from rg_hip import nn as rnn, netgraph as NG
    from rg_hip.tape import RGModule

    class Net(RGModule):
        def __init__(self, p):
            super(Net, self).__init__()
            self.c1 = rnn.Conv2d(8, 8, 3, 1, 1)
            self.drop = rnn.Dropout(p)
            self.c2 = rnn.Conv2d(8, 4, 3, 1, 1)

        def tf(self, tape, x):
            return self.c2.tf(tape, self.drop.tf(tape, self.c1.tf(tape, x, act=rnn.ACT_RELU)))

        def tb(self, tape, dy, need_dx=True):
            return self.c1.tb(tape, self.drop.tb(tape, self.c2.tb(tape, dy)), need_dx=need_dx)

    torch.manual_seed(0)
    x = torch.randn(2, 8, 16, 8, device=dev)
    nd = Net(0.3).to(dev)
    nd.__dict__["_rg_graph"] = True
    for _ in range(5):
        nd(x).sum().backward()
    assert all(rec == 0 and not ok for _, rec, ok in NG.stats(nd).values())

    ng, ne = Net(0.0).to(dev), Net(0.0).to(dev)
    ne.load_state_dict(ng.state_dict())
    ng.__dict__["_rg_graph"] = True
    x2 = torch.randn(2, 8, 16, 8, device=dev)
    for it in range(5):
        for net in (ng, ne):
            net.zero_grad()
            y1, y2 = net(x), net(x2)                      # two forwards in flight
            (y1.sum() + 2.0 * y2.pow(2).sum()).backward()
        for a, b in zip(ng.parameters(), ne.parameters()):
            assert torch.equal(a.grad, b.grad), it
    assert max(rec for _, rec, _ in NG.stats(ng).values()) == 2


def test_uncapturable_network_falls_back_to_eager_with_fresh_caches(dev):
    """A program that cannot be captured (host read-back in tf) runs eagerly — and the capture attempt leaves nothing behind: the
    (r,s)-major filter copies are rebuilt for real (their cache keys are not stamped inside a capture) and the BatchNorm step
    counters are restored.  Compared step by step with a twin that never tries to capture."""
    import warnings
    from rg_hip import nn as rnn, netgraph as NG
    from rg_hip.tape import RGModule

    class Net(RGModule):
        def __init__(self):
            super(Net, self).__init__()
            self.c1 = rnn.Conv2d(16, 16, 3, 1, 1)
            self.bn = rnn.BatchNorm2d(16)
            self.c2 = rnn.Conv2d(16, 8, 3, 1, 1)

        def tf(self, tape, x):
            h = self.bn.tf(tape, self.c1.tf(tape, x), act=rnn.ACT_RELU)
            float(h[0, 0, 0, 0])                          # host read-back: illegal inside a stream capture
            return self.c2.tf(tape, h)

        def tb(self, tape, dy, need_dx=True):
            return self.c1.tb(tape, self.bn.tb(tape, self.c2.tb(tape, dy)), need_dx=need_dx)

    torch.manual_seed(3)
    ng, ne = Net().to(dev), Net().to(dev)
    ne.load_state_dict(ng.state_dict())
    ng.__dict__["_rg_graph"] = True
    og, oe = torch.optim.SGD(ng.parameters(), lr=0.01), torch.optim.SGD(ne.parameters(), lr=0.01)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for it in range(NG.WARMUP + 3):
            x = torch.randn(4, 16, 16, 8, device=dev, generator=torch.Generator(device=dev).manual_seed(50 + it))
            outs = []
            for net, opt in ((ng, og), (ne, oe)):
                opt.zero_grad()
                y = net(x)
                y.pow(2).mean().backward()
                opt.step()
                outs.append(y.detach())
            assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1]), it
            for a, b in zip(ng.parameters(), ne.parameters()):
                assert torch.equal(a, b), it
    assert any("not capturable" in str(w.message) for w in caught)
    assert all(rec == 0 and not ok for _, rec, ok in NG.stats(ng).values())
    assert int(ng.bn.num_batches_tracked) == int(ne.bn.num_batches_tracked)


def test_uncapturable_backward_falls_back_to_eager(dev):
    """forward program captured, backward program not capturable (host read-back in tb): that backward runs eagerly over the record's
    tape, the record is retired and the key stays eager — same numbers as a twin that never captures"""
    import warnings
    from rg_hip import nn as rnn, netgraph as NG
    from rg_hip.tape import RGModule

    class Net(RGModule):
        def __init__(self):
            super(Net, self).__init__()
            self.c1 = rnn.Conv2d(16, 16, 3, 1, 1)
            self.c2 = rnn.Conv2d(16, 8, 3, 1, 1)

        def tf(self, tape, x):
            return self.c2.tf(tape, self.c1.tf(tape, x, act=rnn.ACT_RELU))

        def tb(self, tape, dy, need_dx=True):
            float(dy[0, 0, 0, 0])                         # host read-back: illegal inside a stream capture
            return self.c1.tb(tape, self.c2.tb(tape, dy), need_dx=need_dx)

    torch.manual_seed(5)
    ng, ne = Net().to(dev), Net().to(dev)
    ne.load_state_dict(ng.state_dict())
    ng.__dict__["_rg_graph"] = True
    og, oe = torch.optim.SGD(ng.parameters(), lr=0.01), torch.optim.SGD(ne.parameters(), lr=0.01)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for it in range(NG.WARMUP + 4):
            x = torch.randn(4, 16, 16, 8, device=dev, generator=torch.Generator(device=dev).manual_seed(70 + it)).requires_grad_(True)
            x2 = x.detach().clone().requires_grad_(True)
            for net, opt, xi in ((ng, og, x), (ne, oe, x2)):
                opt.zero_grad()
                net(xi).pow(2).mean().backward()
                opt.step()
            assert torch.equal(x.grad, x2.grad), it
            for a, b in zip(ng.parameters(), ne.parameters()):
                assert torch.isfinite(a).all() and torch.equal(a, b), it
    assert any("backward program" in str(w.message) and "not capturable" in str(w.message) for w in caught)
    assert all(not ok for _, _, ok in NG.stats(ng).values())


def test_capture_survives_event_polls_from_another_thread(dev):
    """The process group's watchdog thread polls the events of in-flight collectives (hipEventQuery) while a network's program may be
    captured; under the default global capture mode such a call from ANY thread invalidates the capture (round 4: the single-rank
    RCCL run of the joint step died that way).  The programs are captured in thread-local mode: a thread hammering Event.query()
    through the whole run changes nothing, and every program is captured and replayed."""
    import threading
    from rg_hip import nn as rnn, netgraph as NG
    from rg_hip.tape import RGModule

    class Net(RGModule):
        def __init__(self):
            super(Net, self).__init__()
            self.c1 = rnn.Conv2d(16, 16, 3, 1, 1)
            self.bn = rnn.BatchNorm2d(16)
            self.c2 = rnn.Conv2d(16, 8, 3, 1, 1)

        def tf(self, tape, x):
            return self.c2.tf(tape, self.bn.tf(tape, self.c1.tf(tape, x), act=rnn.ACT_RELU))

        def tb(self, tape, dy, need_dx=True):
            return self.c1.tb(tape, self.bn.tb(tape, self.c2.tb(tape, dy)), need_dx=need_dx)

    assert NG._CAPTURE_MODE == "thread_local"
    torch.manual_seed(5)
    ng, ne = Net().to(dev), Net().to(dev)
    ne.load_state_dict(ng.state_dict())
    ng.__dict__["_rg_graph"] = True
    og, oe = torch.optim.SGD(ng.parameters(), lr=0.01), torch.optim.SGD(ne.parameters(), lr=0.01)
    stop, polls = threading.Event(), [0]
    side = torch.cuda.Stream(device=dev)

    def poll():
        torch.cuda.set_device(dev)
        buf = torch.zeros(1 << 16, device=dev)
        while not stop.is_set():
            with torch.cuda.stream(side):
                buf.add_(1.0)
                ev = torch.cuda.Event()
                ev.record(side)
            ev.query()
            polls[0] += 1

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    try:
        for it in range(NG.WARMUP + 4):
            x = torch.randn(4, 16, 16, 8, device=dev, generator=torch.Generator(device=dev).manual_seed(70 + it))
            outs = []
            for net, opt in ((ng, og), (ne, oe)):
                opt.zero_grad()
                y = net(x)
                y.pow(2).mean().backward()
                opt.step()
                outs.append(y.detach())
            assert torch.equal(outs[0], outs[1]), it
    finally:
        stop.set()
        th.join(10.0)
    torch.cuda.synchronize()
    assert polls[0] > 0
    for a, b in zip(ng.parameters(), ne.parameters()):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    st = NG.stats(ng)
    assert st and all(ok and records >= 1 for (calls, records, ok) in st.values()), st
