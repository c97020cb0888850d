"""Flow maps that are still on the network's side stream when the model returns them (the literal train_flow.py loop).

`train.Trainer` overlaps the decoder half of pass t (+ the loss container's update()) with the encoders of pass t + 1 by
keeping the flows on a side stream until the window's loss is evaluated.  A caller that runs the reference's own loop body
— ``x = model(inp); x["flow"][i] = x["flow"][i] * flow_scaling; loss.update(x["flow"], ...)`` (train_flow.py:101-118) —
multiplies the flows on ITS stream, so the model had to join the streams before returning and the window ran as on one
stream (36.7 instead of 33 ms).  `LazyFlow` lets that caller keep the overlap without changing a line:

  * the model returns its flows as `LazyFlow` tensors (a `torch.Tensor` subclass; same data, same autograd graph) WITHOUT
    making the caller's stream wait;
  * ``lazy * number`` (either side, `torch.mul`, `.mul`) runs on the side stream and gives a `LazyFlow` again;
  * this package's loss containers take the plain tensor out (`plain_of`), run their update() on the side stream and make
    the evaluation wait for it;
  * EVERYTHING else that touches a `LazyFlow` through torch — any function, method or attribute that goes through
    ``__torch_function__`` — first makes the current stream wait for the side stream and then works on the plain tensor:
    to any other consumer the flows behave like joined flows.

That includes the interfaces that hand the memory to code torch does not see — ``data_ptr()``, DLPack
(``torch.utils.dlpack.to_dlpack`` / ``__dlpack__``), ``__cuda_array_interface__``: on this torch they are dispatched through
``__torch_function__`` like any other method, so the caller's stream has been made to wait before the pointer leaves
(tests/test_lazy_flow_gpu.py::test_raw_pointer_interfaces_join_first, with the decoder half held back).  What the mechanism
cannot cover is foreign code that then uses the pointer on a stream of ITS OWN without ordering it behind the caller's — the
same contract as for any torch tensor.  `TEF_LAZY_FLOWS=0` switches the mechanism off.
"""
import torch

_MUL = {torch.mul, torch.Tensor.mul, torch.Tensor.__mul__, torch.Tensor.__rmul__, torch.multiply, torch.Tensor.multiply}


def plain_of(t):
    """-> (plain tensor, side stream or None).  No torch function is triggered."""
    if type(t) is LazyFlow:
        d = t.__dict__
        return d["_tef_plain"], d["_tef_side"]
    return t, None


class LazyFlow(torch.Tensor):
    @staticmethod
    def wrap(plain, side):
        t = plain.as_subclass(LazyFlow)
        t.__dict__["_tef_plain"] = plain
        t.__dict__["_tef_side"] = side
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _MUL and len(args) == 2 and not kwargs:
            a, b = args
            lazy, other = (a, b) if type(a) is LazyFlow else (b, a)
            if type(lazy) is LazyFlow and isinstance(other, (int, float)) and not isinstance(other, bool):
                plain, side = plain_of(lazy)
                with torch._C.DisableTorchFunctionSubclass():
                    with torch.cuda.stream(side):
                        out = plain * other
                return LazyFlow.wrap(out, side)
        main = None
        plains = []

        def strip(x):
            nonlocal main
            if type(x) is LazyFlow:
                plain, side = plain_of(x)
                if main is None:
                    main = torch.cuda.current_stream(plain.device)
                main.wait_stream(side)
                plain.record_stream(main)
                plains.append(plain)
                return plain
            if isinstance(x, (list, tuple)):
                return type(x)(strip(y) for y in x)
            if isinstance(x, dict):
                return {k: strip(v) for k, v in x.items()}
            return x

        args, kwargs = strip(args), strip(kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)
