"""Exhaustive check of the 99-exchange median-of-25 selection network used by csrc/depth_fill.hip (N. Devillard's opt_med25):
by the 0-1 principle a comparison network selects the median of every input iff it does so on all 2^25 binary inputs.
    python tools/verify_median25.py          (about a minute of numpy)"""
import re
import os
import numpy as np

src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ist-net_amd", "csrc", "depth_fill.hip")).read()
body = src[src.index("float median25(float (&v)[25])"):src.index("#undef ISTNET_CE")]
pairs = np.array(re.findall(r"ISTNET_CE\((\d+), (\d+)\)", body), dtype=int)      # the network as compiled
assert len(pairs) == 99, len(pairs)
chunk = 1 << 22
for start in range(0, 1 << 25, chunk):
    x = np.arange(start, start + chunk, dtype=np.uint32)
    bits = [((x >> i) & 1).astype(np.uint8) for i in range(25)]
    want = (sum(bits) >= 13).astype(np.uint8)
    for a, b in pairs:
        lo, hi = np.minimum(bits[a], bits[b]), np.maximum(bits[a], bits[b])
        bits[a], bits[b] = lo, hi
    assert np.array_equal(bits[12], want), start
print("median25: 99 exchanges, correct on all 2^25 binary inputs")
