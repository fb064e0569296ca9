"""Writes tests/golden/jpeg/vectors.npz: a few JPEG files (bytes) with the pixels Pillow's libjpeg-turbo decodes them to - the
known answers that pin oracle/jpeg_baseline.py where Pillow is not installed.  Run here (Pillow 12.2.0, libjpeg-turbo):
    python tests/golden/make_jpeg_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import jpeg_cases  # noqa: E402

import PIL  # noqa: E402
from PIL import features  # noqa: E402

out = {}
keep = ("37x53_s2_q85_smooth", "37x53_s1_q30_noise", "33x31_s0_q100_smooth", "17x2_s2_q85_smooth", "9x4_s1_q85_noise", "1x1_s2_q85_smooth",
        "37x53_s2_q90_prog", "33x31_s1_q30_prog", "17x2_s2_q100_prog", "41x57_prog_rst", "20x33_gray_prog", "37x53_s2_q85_nonint",
        "64x48_s2_q100_noise", "41x57_qual90_rest3_subs2", "41x57_qual90_rest1_subs1", "24x40_opti", "20x33_gray", "40x56_opti")
names = []
for name, data in jpeg_cases.cases():
    if any(name.startswith(k) for k in keep):
        i = len(names)
        names.append(name)
        out["jpg%d" % i] = np.frombuffer(data, dtype=np.uint8)
        out["rgb%d" % i] = jpeg_cases.pil_decode(data)
out["names"] = np.array(names)
out["made_with"] = np.array("Pillow %s, libjpeg %s, libjpeg_turbo %s" % (PIL.__version__, features.version("jpg"), features.check_feature("libjpeg_turbo")))
os.makedirs(os.path.join(HERE, "jpeg"), exist_ok=True)
np.savez_compressed(os.path.join(HERE, "jpeg", "vectors.npz"), **out)
print(len(names), "vectors:", names)
