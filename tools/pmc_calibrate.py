#!/usr/bin/env python3
"""Known-byte-count launches for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on this library's access
pattern: copy_kernel moves N floats (4 B per lane) -> reads 4N bytes, writes 4N bytes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parrot_tts_amd import _lib  # noqa: E402

n = 256 * 1024 * 1024  # 1 GiB each way: larger than the 256 MiB Infinity Cache
a = torch.randn(n, device="cuda:0")
b = torch.empty_like(a)
for _ in range(3):
    _lib.check(_lib.lib().parrot_debug_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
print("copied", n * 4, "bytes x3")
