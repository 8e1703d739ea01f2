"""Scalar schedule used for the coarse-to-fine windows and the depth-loss epsilon.

API contract of the reference's ``engine/generic_scheduler.py:4-30`` (``update(step)``, the public ``value``
attribute the models read, ``get_value()`` which answers with the end value outside training): a piecewise-linear
ramp that holds ``init_value`` before ``begin_step`` and ``final_value`` after ``end_step``.  Pinned by
``tests/test_glue_cpu.py::test_scheduler_and_chunker_match_reference`` against value tables of the reference class.
"""
from torch import nn


def ramp(step: float, begin_step: float, end_step: float, init_value: float, final_value: float) -> float:
    """Value of the schedule at ``step`` (clamped linear interpolation between the two end points)."""
    if step < begin_step:
        return init_value
    if step > end_step:
        return final_value
    span = end_step - begin_step
    t = (step - begin_step) / span
    t = 0 if t < 0 else (1 if t > 1 else t)
    return init_value + t * (final_value - init_value)


class GenericScheduler(nn.Module):
    """Holds the current value of one schedule; an ``nn.Module`` only so that ``model.train()/eval()`` reaches it."""

    def __init__(self, init_value, final_value, begin_step, end_step) -> None:
        super().__init__()
        self.begin_step, self.end_step = begin_step, end_step
        self.init_value, self.final_value = init_value, final_value
        # a freshly built (or freshly loaded) model behaves like a finished schedule until the first update()
        self.value = final_value

    def update(self, step) -> None:
        self.value = ramp(step, self.begin_step, self.end_step, self.init_value, self.final_value)

    def get_value(self):
        if not self.training:
            return self.final_value
        return self.value
