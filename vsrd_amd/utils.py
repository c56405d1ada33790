"""The few ``vsrd.utils`` helpers scripts/main.py applies to objects of the hot path (reference: vsrd/utils.py)."""
import functools


class Composition:
    """vsrd.utils.compose(f, g, ...): callable x -> ...g(f(x)).  Kept introspectable so that
    ``compose(soft_distance_field, operator.itemgetter(0))`` (main.py:1030) can still be flattened to a field block."""

    def __init__(self, functions):
        self.functions = list(functions)

    def __call__(self, *args, **kwargs):
        first, *rest = self.functions
        value = first(*args, **kwargs)
        for function in rest:
            value = function(value)
        return value


def compose(*functions):
    return Composition(functions)


def reversed_pad(inputs, padding, **kwargs):
    """vsrd/utils.py ``reversed_pad``: like F.pad but the (before, after) pairs are listed from the FIRST dimension on
    (main.py:217-247 pads one zero row after the instances: ``reversed_pad(x, (0, 1))``)."""
    import torch.nn.functional as F
    pairs = [tuple(padding[k:k + 2]) for k in range(0, len(padding), 2)]
    pairs += [(0, 0)] * (inputs.dim() - len(pairs))
    flat = [v for pair in reversed(pairs) for v in pair]
    return F.pad(inputs, flat, **kwargs)
