"""margin_amd -- MI355X-native engine for margin's stRPHmm forward/backward hot path.

The product is the C-ABI shared library ``libmargin_rphmm.so`` (``margin_amd/csrc``, declared in
``include/margin_rphmm.h``); this package is its thin Python binding plus the synthetic chunk
generator used by the tests and the benchmark.
"""
__version__ = "0.1.0"
