"""CPU test of train.WindowRunner's pass bookkeeping (reference train_flow.py:83-87: a new sequence at ANY pass resets the
window) with a stand-in for the captured window: which passes end up in which replay, and when the states are cleared."""
import torch


class _FakeWindow:
    def __init__(self, P):
        self.inputs = [{"x": torch.zeros(1)} for _ in range(P)]
        self.states = [torch.ones(2)]
        self.replays = []

    def replay(self, new_seq=False, exchange=True):
        assert exchange is False and new_seq is False      # (the runner has exchanged the flag pass by pass and cleared the states)
        self.replays.append(([float(d["x"].item()) for d in self.inputs], float(self.states[0].sum().item())))
        self.states[0].fill_(1.0)                           # a window leaves a non-zero recurrent state behind


def test_runner_restarts_the_window_at_the_pass_that_announces_a_new_sequence():
    from taming_event_flow_amd import train

    win = _FakeWindow(3)
    runner = train.WindowRunner(win)
    flags = [False, False, False, False, False, True, False, False, False, True, False, False, True]
    done = [runner.step({"x": torch.tensor([float(t)])}, new_seq=f) for t, f in enumerate(flags)]
    assert done == [False, False, True, False, False, False, False, True, False, False, False, True, False]
    # window 1: passes 0-2 on the carried state; passes 3-4 are dropped by the reset at pass 5; window 2: 5-7 from cleared
    # states; pass 8 dropped by the reset at pass 9; window 3: 9-11 cleared; pass 12 announces again and waits for more
    assert win.replays == [([0.0, 1.0, 2.0], 2.0), ([5.0, 6.0, 7.0], 0.0), ([9.0, 10.0, 11.0], 0.0)]
    assert runner.count == 1 and runner.pending_reset is True
