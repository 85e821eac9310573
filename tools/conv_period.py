"""ConvFusedKernel alone: microseconds per update of a 65 536-tap mono response (BASELINE config 5's slot) on an
otherwise idle GPU, and the rate at which the MAC stage moves its filter + input spectra (2 x 511 segments x 1 KiB).
    python tools/conv_period.py [taps] [updates]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openal-soft_amd"))
import oalgpu

taps = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
updates = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.default_rng(3)
ir = (rng.uniform(-1, 1, taps) * np.exp(-np.arange(taps) / (taps / 5.0)) * 0.05).astype(np.float32)
conv = oalgpu.Convolution(4, ir)
conv.set_target_gains([1.0, 0.5, 0.5, 0.5])
lib = oalgpu.lib
lib.oalgpu_convolution_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
wet = torch.rand(1024, device="cuda") - 0.5
out = torch.zeros(4, 1024, device="cuda")
stream = torch.cuda.current_stream()
def run(k):
    for _ in range(k):
        rc = lib.oalgpu_convolution_process_device(conv.h, C.c_void_p(stream.cuda_stream), C.c_void_p(wet.data_ptr()),
                                                   C.c_void_p(out.data_ptr()), 1024)
        assert rc == 0
run(50)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
run(updates)
e1.record(stream)
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000.0 / updates
segs = taps // 128 - 1
mb = 2 * segs * 1024 / 1e6
print(f"{taps} taps: {us:.1f} us per 1024-sample update; MAC stage operands {mb:.2f} MB -> {mb / us * 1e3:.0f} GB/s effective over the whole update")
