"""plan.StepPlan (train_step._MULTI_GRAPH's engine) on the MI355X: recorded stream topology == eager topology."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def test_join_followed_directly_by_fork_keeps_transitive_order():
    """A.wait_stream(B); C.wait_stream(A) with NO launch on A in between: C's work must still be ordered behind B's (as eagerly and
    under capture).  B runs a long chain of accumulating copies into `acc`; C then reads `acc`: without the transitive edge the
    replay reads it early."""
    from mmego_amd import ops
    from mmego_amd.plan import StepPlan
    dev = _dev()
    n = 1 << 22
    one = torch.ones(n // 64, 64, device=dev)
    acc = torch.zeros(n // 64, 64, device=dev)
    out = torch.zeros(n // 64, 64, device=dev)
    sb, sc = torch.cuda.Stream(), torch.cuda.Stream()

    def body():
        a = torch.cuda.current_stream()
        ops.fill(acc, 0.0)
        sb.wait_stream(a)
        with torch.cuda.stream(sb):
            for _ in range(40):
                ops.copy2d(one, acc, accumulate=True)
        a.wait_stream(sb)                 # join ...
        sc.wait_stream(a)                 # ... followed directly by a fork: nothing was launched on `a` in between
        with torch.cuda.stream(sc):
            ops.copy2d(acc, out)
        a.wait_stream(sc)
    body()
    torch.cuda.synchronize()
    assert float(out.min()) == 40.0
    plan = StepPlan().record(body).build()
    for how in (plan.run_eagerly, plan.replay, plan.replay):
        out.zero_()
        torch.cuda.synchronize()
        how()
        torch.cuda.synchronize()
        assert float(out.min()) == 40.0 and float(out.max()) == 40.0, how.__name__


def test_record_refuses_what_it_cannot_replay():
    from mmego_amd import ops
    from mmego_amd.plan import StepPlan
    dev = _dev()
    buf = torch.zeros(64, 64, device=dev)
    ev = torch.cuda.Event()
    for bad in (lambda: ev.record(), lambda: torch.cuda.current_stream().wait_event(ev)):
        def body():
            ops.fill(buf, 1.0)
            bad()
        with pytest.raises(RuntimeError, match="only hip.call launches"):
            StepPlan().record(body)
    # the class-wide patches are gone again
    ev.record()
    torch.cuda.current_stream().wait_event(ev)
    torch.cuda.synchronize()


def test_torch_op_in_a_body_is_caught_by_comparing_with_the_eager_body():
    """A device op that bypasses hip.call runs once at record time and is missing from every replay -- nothing can intercept it,
    so the contract is checked the way the engine tests do: replay vs eager body."""
    from mmego_amd import ops
    from mmego_amd.plan import StepPlan
    dev = _dev()
    a = torch.zeros(64, 64, device=dev)
    b = torch.zeros(64, 64, device=dev)

    def body():
        ops.fill(a, 2.0)
        b.copy_(a)                         # deliberate violation: a torch op
        ops.copy2d(b, a, accumulate=True)
    body()
    torch.cuda.synchronize()
    want = a.clone()                       # eager: a = 2 + 2
    plan = StepPlan().record(body).build()
    a.zero_(); b.zero_()
    plan.replay()
    torch.cuda.synchronize()
    assert not torch.equal(a, want), "the missing torch op must show up as a difference"
