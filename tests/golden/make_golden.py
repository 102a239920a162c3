#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- run in the authoring container only (needs /root/reference).

The reference has NO golden vectors, so these are outputs of the CPU oracle
(oracle/rtdd_oracle.c, itself pinned by tests/test_oracle.py) on 256x256 crops of three bundled
image/annotation pairs.  Inputs are stored DECODED (raw u8 arrays) so JPEG/PNG decoder
differences cannot leak in.  Only data is stored here -- no reference source text.

Per crop:
  bgr, gray0, annotation (decoded), mask0/edited0 after the reference's decode rule
  (src/main.cpp:160-168: gray != 32 -> label, mask 255), the cascade of SURVEY.md A.5 with P = 3
  (256/128/64 px, 250/500/1000 sweeps): per-level gray, mask, edited ch0, depth BEFORE and AFTER
  each GPUMatrixFreeSolver call for the contracted variant (f32, exact), sha256 of the same for
  the non-contracted variant plus the max-abs spread between the two, the int2 index maps'
  sha256, the three depth effects on the final depth, and scipy's direct solution of the
  coarsest-level linear system.
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference/dataset"
CROPS = {"Dog": None, "Arara": None, "WomanParasol": None}     # name -> (y0, x0), chosen below and recorded in the file
SIZE = 256
LEVELS = 3
ITERS = [250, 500, 1000]                                        # level 0, 1, 2  (src/main.cpp:263)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pick_crop(ann):
    """Deterministic: the window (stride 16) with the most distinct labels, then the most annotated pixels."""
    best = None
    for y0 in range(0, ann.shape[0] - SIZE + 1, 16):
        for x0 in range(0, ann.shape[1] - SIZE + 1, 16):
            w = ann[y0:y0 + SIZE, x0:x0 + SIZE]
            key = (len(np.unique(w[w != 32])), int((w != 32).sum()))
            if 0.03 < (w != 32).mean() < 0.3 and (best is None or key > best[0]):
                best = (key, y0, x0)
    return best[1], best[2]


def direct_solution(gray, mask, depth, lut):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    rows, cols = gray.shape
    idx = oracle.index_to_weight(gray, None, 0, 0)
    n = rows * cols
    A = sp.lil_matrix((n, n)); b = np.zeros(n)
    for y in range(rows):
        for x in range(cols):
            i = y * cols + x
            if mask[y, x] == 255:
                A[i, i] = 1.0; b[i] = depth[y, x]; continue
            l, r = divmod(int(idx[y, x, 0]), 1000); u, d = divmod(int(idx[y, x, 1]), 1000)
            tot = 0.0
            for k, j in ((l, i - 1), (r, i + 1), (u, i - cols), (d, i + cols)):
                if k != 256:
                    w = float(lut[k]); A[i, j] = -w; tot += w
            A[i, i] = tot if tot > 0 else 1.0
    return spla.spsolve(A.tocsr(), b).reshape(rows, cols).astype(np.float64)


def cascade(gray0, mask0, edited0, lut, contract):
    gray = [gray0]; mask = [mask0.copy()]; edited = [edited0.copy()]
    for lvl in range(1, LEVELS):
        gray.append(oracle.pyrdown_u8(gray[-1]))
        r, c = SIZE >> lvl, SIZE >> lvl
        m = np.zeros((r, c), np.uint8); e = np.zeros((r, c, 3), np.uint8)
        oracle.pyrdown_annotation(mask[-1], edited[-1], m, e)
        mask.append(m); edited.append(e)
    depth = [np.full((SIZE >> l, SIZE >> l), 255.0, np.float32) for l in range(LEVELS)]
    before, after, index_sha = {}, {}, {}
    oracle.convert_to_float(edited[LEVELS - 1], depth[LEVELS - 1], mask[LEVELS - 1])
    for lvl in range(LEVELS - 1, -1, -1):
        before[lvl] = depth[lvl].copy()
        index_sha[lvl] = sha(oracle.index_to_weight(gray[lvl], depth[lvl], lvl, LEVELS - 1))
        oracle.solve(depth[lvl], mask[lvl], gray[lvl], ITERS[lvl], lvl, LEVELS - 1, lut, contract, threads=4)
        after[lvl] = depth[lvl].copy()
        if lvl > 0:
            depth[lvl - 1] = oracle.pyrup_f32(depth[lvl], SIZE >> (lvl - 1), SIZE >> (lvl - 1), contract=contract)
            oracle.convert_to_float(edited[lvl - 1], depth[lvl - 1], mask[lvl - 1])
    return gray, mask, edited, before, after, index_sha


def main():
    oracle.build()
    lut = oracle.load_weights(0.4)
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in CROPS:
        rgb = np.array(Image.open(f"{REF}/images/{name}.jpg").convert("RGB"))
        ann = np.array(Image.open(f"{REF}/annotations/{name}.png").convert("RGB"))[..., 0]
        y0, x0 = pick_crop(ann)
        bgr = np.ascontiguousarray(rgb[y0:y0 + SIZE, x0:x0 + SIZE, ::-1])
        ann = np.ascontiguousarray(ann[y0:y0 + SIZE, x0:x0 + SIZE])
        gray0 = oracle.bgr2gray(bgr)
        mask0 = np.where(ann != 32, 255, ann).astype(np.uint8)               # src/main.cpp:163-166
        edited0 = bgr.copy(); edited0[ann != 32] = ann[ann != 32][:, None]    # B=G=R=label
        data = {"name": name, "crop_y0x0": np.array([y0, x0]), "bgr": bgr, "annotation": ann, "gray0": gray0,
                "mask0": mask0, "edited0": edited0, "lut": lut, "iters": np.array(ITERS)}
        res = {}
        for contract in (1, 0):
            res[contract] = cascade(gray0, mask0, edited0, lut, contract)
        gray, mask, edited, before, after, index_sha = res[1]
        for lvl in range(LEVELS):
            data[f"gray{lvl}"] = gray[lvl]; data[f"mask{lvl}"] = mask[lvl]; data[f"edited_ch0_{lvl}"] = edited[lvl][..., 0].copy()
            data[f"depth_before_c1_L{lvl}"] = before[lvl]; data[f"depth_after_c1_L{lvl}"] = after[lvl]
            data[f"index_sha_c1_L{lvl}"] = index_sha[lvl]
            data[f"depth_after_c0_sha_L{lvl}"] = sha(res[0][4][lvl])
            data[f"depth_before_c0_sha_L{lvl}"] = sha(res[0][3][lvl])
            data[f"spread_c0_c1_L{lvl}"] = np.float64(np.abs(res[0][4][lvl] - after[lvl]).max())
        final = after[0]
        data["depth_u8"] = oracle.depth_to_u8(final)
        data["desaturate_c1"] = oracle.desaturate(bgr, gray0, final, 1)
        data["haze_c1"] = oracle.haze(bgr, final, 1)
        data["defocus"] = oracle.defocus(bgr, final, threads=4)
        lo = np.ascontiguousarray(gray[2] >> 4)                               # well-conditioned variant for the converged check
        data["direct_gray_L2"] = lo
        data["direct_solution_L2"] = direct_solution(lo, mask[2], before[2], lut)
        np.savez_compressed(os.path.join(out_dir, f"{name}_256.npz"), **data)
        print(name, "crop", (y0, x0), "labels", np.unique(ann[ann != 32]), "coverage %.3f" % (ann != 32).mean(),
              "spread c0/c1 per level", [float(data[f"spread_c0_c1_L{l}"]) for l in range(LEVELS)])


def make_full(name="Dog"):
    """One FULL-SIZE bundled pair, decoded (inputs only + hashes of the oracle's outputs): the end-to-end parity test at the
    dataset's own resolution computes the expected images with the oracle at test time."""
    from cascade_ref import Cascade
    oracle.build()
    lut = oracle.load_weights(0.4)
    rgb = np.array(Image.open(f"{REF}/images/{name}.jpg").convert("RGB"))
    ann = np.array(Image.open(f"{REF}/annotations/{name}.png").convert("RGB"))[..., 0]
    bgr = np.ascontiguousarray(rgb[..., ::-1])
    c = Cascade(oracle, bgr, ann, lut, 1, threads=8)
    c.estimate(1000)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{name}_full.npz"), name=name, bgr=bgr, annotation=np.ascontiguousarray(ann),
                        levels=np.array(c.P), depth_u8_sha=sha(c.depth_u8), depth_sha=sha(c.depth[0]))
    print(name, "full size", bgr.shape, "levels", c.P, "labels", np.unique(ann[ann != 32]))


def make_dataset():
    """ALL twelve bundled pairs at their own resolution (src/main.cpp:93-113,160-173,232-295 run on dataset/images +
    dataset/annotations).  Stored DECODED and lossless -- tests/golden/dataset/<name>.png (RGB pixels as the JPEG decoder of
    the authoring container produced them) and <name>_ann.png (channel 0 of the annotation; R = G = B in all twelve) -- so
    JPEG decoder differences cannot leak in.  Inputs only; of the oracle's outputs the manifest keeps sha256 hashes (every
    level's depth in both contraction variants, the u8 map, the three effects), which the GPU test re-derives on the box."""
    import json
    from cascade_ref import Cascade
    oracle.build()
    lut = oracle.load_weights(0.4)
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataset")
    os.makedirs(out_dir, exist_ok=True)
    manifest = {}
    for f in sorted(os.listdir(f"{REF}/images")):
        if not f.endswith(".jpg"):
            continue
        name = f[:-4]
        rgb = np.array(Image.open(f"{REF}/images/{name}.jpg").convert("RGB"))
        ann3 = np.array(Image.open(f"{REF}/annotations/{name}.png").convert("RGB"))
        assert (ann3[..., 0] == ann3[..., 1]).all() and (ann3[..., 0] == ann3[..., 2]).all() and ann3.shape == rgb.shape
        ann = np.ascontiguousarray(ann3[..., 0])
        Image.fromarray(rgb, "RGB").save(os.path.join(out_dir, f"{name}.png"), optimize=True)
        Image.fromarray(ann, "L").save(os.path.join(out_dir, f"{name}_ann.png"), optimize=True)
        bgr = np.ascontiguousarray(rgb[..., ::-1])
        entry = {"rows": int(rgb.shape[0]), "cols": int(rgb.shape[1]), "labels": [int(v) for v in np.unique(ann[ann != 32])],
                 "coverage": float((ann != 32).mean()), "rgb_sha": sha(rgb), "annotation_sha": sha(ann)}
        for contract in (1, 0):
            c = Cascade(oracle, bgr, ann, lut, contract, threads=8)
            c.estimate(1000)
            entry["levels"] = c.P
            entry["sizes"] = [list(s) for s in c.size]
            entry["gray_sizes"] = [list(g.shape) for g in c.gray]
            entry[f"depth_sha_c{contract}"] = [sha(c.depth[l]) for l in range(c.P)]
            entry[f"depth_u8_sha_c{contract}"] = sha(c.depth_u8)
            if contract == 1:
                entry["gray_sha"] = [sha(g) for g in c.gray]
                entry["desaturate_sha"] = sha(oracle.desaturate(bgr, c.gray[0], c.depth[0], 1))
                entry["haze_sha"] = sha(oracle.haze(bgr, c.depth[0], 1))
                entry["defocus_sha"] = sha(oracle.defocus(bgr, c.depth[0], threads=8))
                first = c.depth[0].copy()
            else:
                entry["spread_c0_c1_level0"] = float(np.abs(first - c.depth[0]).max())
        manifest[name] = entry
        print(name, entry["rows"], entry["cols"], "levels", entry["levels"], "sizes", entry["sizes"], "gray", entry["gray_sizes"])
    with open(os.path.join(out_dir, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)


def copy_dataset_jpegs():
    """The twelve photographs AS FILES (dataset/images/<name>.jpg, byte for byte) beside their decoded copies: what the harness's own
    JPEG reader (harness/jpeg_reader.hpp) is checked on -- its pixels must equal <name>.png, i.e. libjpeg-turbo's (tests/test_harness_jpeg.py)."""
    import shutil
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataset")
    for f in sorted(os.listdir(f"{REF}/images")):
        if f.endswith(".jpg"):
            rgb = np.array(Image.open(f"{REF}/images/{f}").convert("RGB"))
            assert np.array_equal(rgb, np.array(Image.open(os.path.join(out_dir, f[:-4] + ".png")).convert("RGB"))), f
            shutil.copyfile(f"{REF}/images/{f}", os.path.join(out_dir, f))
            os.chmod(os.path.join(out_dir, f), 0o644)
            print("copied", f)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--dataset-jpeg":
        copy_dataset_jpegs()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--dataset":
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        make_dataset()
    elif len(sys.argv) > 1 and sys.argv[1] == "--full":
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        make_full(*sys.argv[2:])
    else:
        main()
