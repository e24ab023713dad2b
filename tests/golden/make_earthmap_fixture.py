"""Regenerates path_tracer_amd/data/earthmap_rgb8.png from the reference's earthmap.jpg.

The reference decodes earthmap.jpg with image 0.23.9 / jpeg-decoder 0.1.20
(src/texture/picture.rs:11, Cargo.lock:137-162); that decoder is not available
here, so texel values are "parity unpinned" (SURVEY.md section 8(c)).  This script
decodes the JPEG ONCE with PIL/libjpeg-turbo and stores the decoded RGB8 texels
losslessly, so every later run (tests, smoke, bench, CPU container and GPU box)
feeds oracle and GPU the same bytes regardless of the JPEG decoder installed.
Run in the build container only (needs /root/reference).
"""
import hashlib
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/earthmap.jpg"
dst = sys.argv[2] if len(sys.argv) > 2 else "path_tracer_amd/data/earthmap_rgb8.png"
im = Image.open(src).convert("RGB")
a = np.asarray(im, dtype=np.uint8)
Image.fromarray(a).save(dst, format="PNG", optimize=True)
back = np.asarray(Image.open(dst).convert("RGB"), dtype=np.uint8)
assert (back == a).all()
print(a.shape, "sha256(rgb8) =", hashlib.sha256(a.tobytes()).hexdigest())
