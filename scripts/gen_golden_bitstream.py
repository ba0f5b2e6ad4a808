#!/usr/bin/env python3
"""Golden vectors of the first partition (frame header + macroblock modes / motion vectors), produced by the
REFERENCE's own encode_header (src/entropy_host.cpp, compiled by oracle/build_ref.sh into
oracle/_ref/libvp8refhost.so).  Runs only where /root/reference exists; tests/golden/bitstream/*.npz hold inputs and
the expected bytes, no code.

    python scripts/gen_golden_bitstream.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bitstream_cases import default_sd, random_inter_case, ref_encode_header, ref_header_lib  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "bitstream")


def main():
    if ref_header_lib() is None:
        raise SystemExit("oracle/_ref/libvp8refhost.so is not built (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    cases = {"inter_11x9_mixed": (11, 9, 21, dict(), (0, 0, 0), 2, 1), "inter_7x5_split_long": (7, 5, 22, dict(split=1.0, long_mv=0.5), (0, 1, 0), 0, 0),
             "inter_9x6_intra20": (9, 6, 23, dict(intra=0.2), (0, 0, 1), 7, 3), "inter_9x6_intra2": (9, 6, 24, dict(intra=0.02), (0, 0, 0), 3, 2),
             "key_8x6": (8, 6, 25, dict(), (1, 1, 1), 4, 0)}
    for name, (mbw, mbh, seed, kw, flags, sharp, plog) in cases.items():
        c = random_inter_case(mbw, mbh, seed, **kw)
        key = flags[0] == 1
        sd = default_sd(key)
        if key:
            c.update(is_inter=None, replaced=0)
        args = dict(ref_frame=None if key else c["ref_frame"], parts=None if key else c["parts"], vectors=None if key else c["vectors"],
                    is_inter=c["is_inter"], modes=c["modes"], replaced=c["replaced"], sharpness=sharp, partitions_log2=plog)
        hdr = ref_encode_header(mbw * 16, mbh * 16, flags, sd, c["seg"], c["nz"], c["probs"], c["denom"], c["skip_prob"], **args)
        d = dict(W=mbw * 16, H=mbh * 16, flags=np.asarray(flags), sd=sd, seg=c["seg"], nz=c["nz"], probs=c["probs"], denom=c["denom"],
                 skip_prob=c["skip_prob"], replaced=c["replaced"], sharpness=sharp, partitions_log2=plog, modes=c["modes"], header=hdr)
        if not key:
            d.update(ref_frame=c["ref_frame"], parts=c["parts"], vectors=c["vectors"], is_inter=c["is_inter"])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, len(hdr), "bytes")


if __name__ == "__main__":
    main()
