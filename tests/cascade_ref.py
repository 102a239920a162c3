"""The depth-estimate loop of /root/reference/src/main.cpp:92-155 + 232-295, composed from oracle
functions (test infrastructure).  Returns every intermediate so tests can localise a mismatch."""
import math

import numpy as np


def pyramid_levels(rows, cols):
    return int(math.log2(max(min(cols, rows) // 45, 1)) + 1)            # main.cpp:95


class Cascade:
    def __init__(self, oracle, bgr, annotation, lut, contract=1, threads=1):
        self.o = oracle; self.lut = lut; self.contract = contract; self.threads = threads
        rows, cols = bgr.shape[:2]
        self.P = P = pyramid_levels(rows, cols)
        self.size = [(int(np.float32(rows) / np.float32(2.0) ** l), int(np.float32(cols) / np.float32(2.0) ** l)) for l in range(P)]   # :103
        self.gray = [oracle.bgr2gray(bgr)]
        for l in range(1, P):
            self.gray.append(oracle.pyrdown_u8(self.gray[-1]))           # ceil chain
        self.scribble = [np.zeros(s, np.uint8) for s in self.size]
        self.edited = [np.zeros(s + (3,), np.uint8) for s in self.size]
        self.depth = [np.full(s, 255.0, np.float32) for s in self.size]  # :136
        self.edited[0][...] = bgr                                        # :158
        if annotation is not None:                                       # :160-168
            lab = annotation != 32
            self.edited[0][lab] = annotation[lab][:, None]
            self.scribble[0] = np.where(lab, 255, annotation).astype(np.uint8)

    def estimate(self, max_iterations=1000):
        o, P = self.o, self.P
        for l in range(1, P):                                            # :249-253
            o.pyrdown_annotation(self.scribble[l - 1], self.edited[l - 1], self.scribble[l], self.edited[l])
        o.convert_to_float(self.edited[P - 1], self.depth[P - 1], self.scribble[P - 1])   # :257
        self.before = {}
        for l in range(P - 1, -1, -1):
            iters = int(np.float32(max_iterations) / np.float32(2.0) ** ((P - 1) - l))    # :263
            r, c = self.size[l]
            self.before[l] = self.depth[l].copy()
            if r > 0 and c > 0:
                o.solve(self.depth[l], self.scribble[l], self.gray[l], iters, l, P - 1, self.lut, self.contract, threads=self.threads, rows=r, cols=c)
            if l > 0:
                self.depth[l - 1] = o.pyrup_f32(self.depth[l], *self.size[l - 1], contract=self.contract)          # :272-279
                o.convert_to_float(self.edited[l - 1], self.depth[l - 1], self.scribble[l - 1])   # :281
        self.depth_u8 = o.depth_to_u8(self.depth[0])                     # :290
        return self.depth[0]
