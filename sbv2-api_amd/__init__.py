"""MI355X-native replacement for sbv2_core's model-execution path (bert::predict + model::synthesize).

Python here is only the ctypes host mirror of the C ABI in include/sbv2_hip.h (used by tests and bench.py);
the product is libsbv2_hip.so (csrc/).  Nothing in this package imports oracle/.
"""
