"""MI355X-native FreqFusion x4 inference path (hand-written HIP kernels behind a C ABI).

Only what the hot path needs: csrc/ (kernels + C ABI), the ctypes binding and the Python host that
sequences the kernels exactly as the reference's eval forward does.  Import as ``isr2_amd``.
"""
