"""coffeedb_amd — MI355X-native text index for CoffeeDB's string-index hot path.

The product is the C-ABI library built from coffeedb_amd/csrc (see include/coffeedb_gpu.h); this
package only holds the ctypes binding used by tests and bench.py, and the synthetic workloads.
"""
