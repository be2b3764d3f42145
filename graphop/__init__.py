"""Drop-in alias of the reference's extension module name: ``from graphop import *`` binds the
same eight functions the reference's pybind11 module exports (graphop/graphop.cpp:216-225)."""
from custom_op_benchmark_amd.graphop import *  # noqa: F401,F403
from custom_op_benchmark_amd.graphop import __all__  # noqa: F401
