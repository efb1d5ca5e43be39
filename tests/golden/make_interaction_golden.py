#!/usr/bin/env python3
"""Generate tests/golden/tri_interaction.npz from the REFERENCE's own
Triangle::InteractionFromIntersection (oracle/_ref/ref_interaction, built by oracle/Makefile from
/root/reference sources).  Build-container only; the .npz holds data only: seeded inputs and the
reference's outputs as float32 bit patterns."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_interaction")
sys.path.insert(0, HERE)
from interaction_cases import cases, patch_cases, transform_cases  # noqa: E402


def run_ref(rec, mode="tri"):
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(rec)).tobytes())
            f.write(np.ascontiguousarray(rec, np.float32).tobytes())
        subprocess.run([REF, mode, fi, fo], check=True)
        return np.fromfile(fo, dtype=np.uint32).reshape(len(rec), {"tri": 44, "blp": 50, "xf": 40}[mode])


if __name__ == "__main__":
    rec = cases(4096, 20240607)
    out = run_ref(rec)
    np.savez_compressed(os.path.join(HERE, "tri_interaction.npz"), inputs=rec, outputs=out)
    print("tri_interaction.npz:", rec.shape, out.shape)
    rec = patch_cases(3072, 20240608)
    out = run_ref(rec, "blp")
    np.savez_compressed(os.path.join(HERE, "blp_interaction.npz"), inputs=rec, outputs=out)
    print("blp_interaction.npz:", rec.shape, out.shape)
    rec = transform_cases(3072, 20240609)
    out = run_ref(rec, "xf")
    np.savez_compressed(os.path.join(HERE, "xf_interaction.npz"), inputs=rec, outputs=out)
    print("xf_interaction.npz:", rec.shape, out.shape)
