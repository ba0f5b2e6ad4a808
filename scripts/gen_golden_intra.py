#!/usr/bin/env python3
"""Golden vectors of the host intra path, produced by the REFERENCE's own code (src/intra_part.h and check_SSIM of
src/vp8enc.cpp compiled for x86 by oracle/build_ref.sh, driven by oracle/ref_host_driver.cpp).  Runs only where
/root/reference exists; the fixtures it writes (tests/golden/intra/*.npz: inputs + expected outputs, no code)
travel with the repo.

    python scripts/gen_golden_intra.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from intra_cases import CHECK_KEYS, INTRA_KEYS, fallback_case, key_case  # noqa: E402
from oracle_lib import ref_intra  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "intra")


def main():
    ref = ref_intra()
    if ref is None:
        raise SystemExit("oracle/_ref/libvp8refhost.so is not built (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    for name, (W, H, seed, qi, kind) in {"key_64x48_q0": (64, 48, 1, 0, "synth"), "key_96x64_q30": (96, 64, 2, 30, "synth"),
                                         "key_48x32_noise_q10": (48, 32, 3, 10, "noise"), "key_16x16_flat": (16, 16, 4, 0, "flat")}.items():
        cur, sd = key_case(W, H, seed, qi, kind)
        r = ref.intra_transform(cur, sd)
        r["MB_coeffs"][:, 24] = 0
        np.savez_compressed(os.path.join(OUT, name + ".npz"), cur_Y=cur[0], cur_U=cur[1], cur_V=cur[2], sd=sd,
                            **{"out_" + k: r[k] for k in INTRA_KEYS})
        print(name, "modes", np.bincount(r["modes"].ravel(), minlength=10).tolist())
    for name, (W, H, seed, target, cut, qi) in {"check_96x64_cut_t97_q40": (96, 64, 5, 0.97, True, (40, 100)),
                                                "check_96x64_cut_t90_q60": (96, 64, 5, 0.90, True, (60, 120)),
                                                "check_64x64_t995": (64, 64, 6, 0.995, False, (0, 48)),
                                                "check_112x80_cut_t98_q20": (112, 80, 7, 0.98, True, (20, 80))}.items():
        cur, sd, inter = fallback_case(W, H, seed, target, scene_cut=cut, qi=qi)
        r = ref.check_ssim(cur, sd, target, inter)
        repl = r["is_inter"] == 0
        r["MB_coeffs"][repl, 24] = 0          # undefined in the reference (uninitialised stack copy)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), cur_Y=cur[0], cur_U=cur[1], cur_V=cur[2], sd=sd, target=np.float32(target),
                            **{"in_" + k: v for k, v in inter.items()}, **{"out_" + k: r[k] for k in CHECK_KEYS},
                            replaced=r["replaced"], new_SSIM=r["new_SSIM"], filter_updated=r["filter_updated"])
        print(name, "below", int((inter["MB_SSIM"] < target).sum()), "replaced", r["replaced"], "new_SSIM", r["new_SSIM"])


if __name__ == "__main__":
    main()
