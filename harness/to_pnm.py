#!/usr/bin/env python3
"""Convert any image Pillow can read (the dataset's JPEGs) to the binary PPM / PGM the harness reads.
usage: to_pnm.py in.jpg out.ppm        (colour -> P6; add --gray for a one-channel P5)
The harness reads JPEG (baseline and progressive Huffman, 8 bit: harness/jpeg_reader.hpp), 8-bit PNG and PNM itself; this is for the rest."""
import sys
from PIL import Image
gray = "--gray" in sys.argv
args = [a for a in sys.argv[1:] if a != "--gray"]
im = Image.open(args[0]).convert("L" if gray else "RGB")
with open(args[1], "wb") as f:
    f.write(b"%s\n%d %d\n255\n" % (b"P5" if gray else b"P6", im.width, im.height))
    f.write(im.tobytes())
