"""Race hunt for the software-pipelined bf16x6 kernels (conv_split.hip): repeated launches over random shapes, including full-machine
ones, each compared with the fp32-MFMA kernel.  A missed barrier shows up as a sporadic large error."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_pipelined_split_kernels_have_no_sporadic_errors():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'split_stress.py')
    spec = importlib.util.spec_from_file_location('split_stress', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    launches, bad = mod.run(40, seed=7)
    assert launches > 300 and bad == 0
