"""GenericScheduler -- mirror of the reference's engine/generic_scheduler.py:4-30: linear ramp of a scalar
between two steps; ``get_value()`` returns the final value in eval mode."""
import torch


class GenericScheduler(torch.nn.Module):
    def __init__(self, init_value, final_value, begin_step, end_step) -> None:
        super().__init__()
        self.init_value, self.final_value = init_value, final_value
        self.begin_step, self.end_step = begin_step, end_step
        self.value = final_value

    def update(self, step) -> None:
        if step > self.end_step:
            self.value = self.final_value
        elif step < self.begin_step:
            self.value = self.init_value
        else:
            frac = min(max((step - self.begin_step) / (self.end_step - self.begin_step), 0), 1)
            self.value = self.init_value + frac * (self.final_value - self.init_value)

    def get_value(self):
        return self.value if self.training else self.final_value
