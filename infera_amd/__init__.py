"""infera_amd -- MI355X-native backend for the Infera in-database inference path.

The product is the shared library ``libinfera.so`` (C ABI: include/infera.h, include/infera_hip.h)
built from ``infera_amd/csrc``.  This Python package only holds the ctypes binding used by tests
and bench.py (``capi``), the ONNX model writer for the benchmark configurations
(``onnx_writer``) and the synthetic-table generator (``synth``).  There is no Python or CPU
implementation of the compute path: without the library and a gfx950 GPU, predictions fail loudly.
"""
__version__ = "0.4.0-mi355x"
