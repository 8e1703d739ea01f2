"""Experiment (round 6): what would the level-parallel rank gain if its table optimizer pass started as soon as the gradient
planes are complete -- right after the backward exchange -- instead of after the deformation backward and the small gradients'
all-reduce?  The pass then runs beside the deformation backward instead of alone at the end of the step's chain.

This is a TIMING stand-in, not a training path: the pass is launched with a zero found_inf flag (the real flag -- the owners'
non-finite flags and the reduced fused-MLP gradients of the same optimizer group -- would need a small collective of its own at
that point), which is harmless here only because `bench.py --level-parallel-one-rank` freezes every learning rate for its
steady state.  usage: python tools/experiments/lp_early_table_step.py <bench.py arguments>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from nersemble_amd.engine.level_parallel import LevelParallel, LevelParallelTableAdam  # noqa: E402

_orig_backward = LevelParallel.backward
_orig_step = LevelParallelTableAdam.step


def _backward(self, *args, **kw):
    out = _orig_backward(self, *args, **kw)
    opt = getattr(self, "_early_opt", None)
    if opt is not None and kw.get("need_table", True) and opt._early_args is not None:
        inv, side = opt._early_args
        if opt._early_zero is None:
            opt._early_zero = torch.zeros((1,), dtype=torch.float32, device=self.he.tables.device)
        _orig_step(opt, found_inf=opt._early_zero, inv_scale=inv, side_stream=side)
        opt._early_done = True
    return out


def _step(self, found_inf=None, inv_scale=None, side_stream=None):
    self.lp._early_opt = self
    self._early_args = (inv_scale, side_stream)
    if getattr(self, "_early_done", False):
        self._early_done = False
        return None
    return _orig_step(self, found_inf=found_inf, inv_scale=inv_scale, side_stream=side_stream)


LevelParallelTableAdam._early_args = None
LevelParallelTableAdam._early_zero = None
LevelParallelTableAdam._early_done = False
LevelParallel.backward = _backward
LevelParallelTableAdam.step = _step

if __name__ == "__main__":
    bench.main()
