#!/usr/bin/env python3
"""Convert any image PIL can read (JPEG, PNG, ...) to the binary PPM the CLI reads.
usage: python tools/img2ppm.py in.jpg out.ppm"""
import sys
from PIL import Image
Image.open(sys.argv[1]).convert("RGB").save(sys.argv[2], format="PPM")
