"""Inputs for the coefficient entropy stage tests: macroblock coefficient sets as the inter path produces them
(short[MBs][25][16] in zig-zag order, MB_parts, MB_non_zero_coeffs) -- real ones from the oracle's inter path and
synthetic stress sets (dense blocks, every token category, skipped macroblocks, mixed 16x16 / 8x8)."""
import numpy as np


def nz_counts(coeffs: np.ndarray, parts: np.ndarray) -> np.ndarray:
    """prepare_filter_mask's count (src/CPU_kernels.cl:800-819): sum |c| over Y AC, all U/V, plus block 24 for
    16x16 macroblocks or the sixteen Y DCs otherwise."""
    a = np.abs(coeffs.astype(np.int64))
    nz = a[:, :16, 1:].sum(axis=(1, 2)) + a[:, 16:24, :].sum(axis=(1, 2))
    nz += np.where(parts == 0, a[:, 24, :].sum(axis=1), a[:, :16, 0].sum(axis=1))
    return nz.astype(np.int32)


def synthetic(mbw: int, mbh: int, seed: int, density: float = 0.25, big: float = 0.02, skip: float = 0.2,
              p16: float = 0.6):
    """Random coefficient sets with a geometric fall-off along the zig-zag scan."""
    rng = np.random.default_rng(seed)
    mbs = mbw * mbh
    fall = density * np.exp(-np.arange(16) / 5.0)
    live = rng.random((mbs, 25, 16)) < fall
    mag = rng.geometric(0.45, size=(mbs, 25, 16))
    bigm = rng.integers(5, 2049, size=(mbs, 25, 16))
    mag = np.where(rng.random((mbs, 25, 16)) < big, bigm, mag)
    sign = np.where(rng.random((mbs, 25, 16)) < 0.5, -1, 1)
    coeffs = (live * mag * sign).astype(np.int16)
    parts = (rng.random(mbs) >= p16).astype(np.int32)          # 0 = 16x16 (has Y2), 1 = 8x8
    coeffs[parts != 0, 24, :] = 0                                # no Y2 block without 16x16 prediction
    dead = rng.random(mbs) < skip
    coeffs[dead] = 0
    # 16x16 macroblocks keep reconstructed DCs in the Y DC slots (never coded, never counted): make them noisy
    dc = rng.integers(-300, 301, size=(mbs, 16)).astype(np.int16)
    for m in np.nonzero(parts == 0)[0]:
        coeffs[m, :16, 0] = dc[m]
    return coeffs, parts, nz_counts(coeffs, parts)


def from_inter_path(W: int, H: int, seed: int, qi=(12, 24, 36, 48)):
    """Coefficients of one real inter frame (oracle path) on the synthetic sequence."""
    from oracle_lib import Oracle
    from pipeline import default_segments
    from vp8oclenc_amd.synth import SynthSequence
    s = SynthSequence(W, H, seed=seed)
    o = Oracle(s.W, s.H)
    o.set_segments(default_segments())
    o.upload_last(*s.frame(0))
    o.upload_current(*s.frame(1))
    o.inter_transform(0, 0, 0, 0)
    r = o.download_results()
    o.loop_filter()
    nz = o.filter_outputs()["MB_non_zero_coeffs"]       # prepare_filter_mask's counts
    o.close()
    return (np.ascontiguousarray(r["MB_coeffs"], np.int16), np.ascontiguousarray(r["MB_parts"], np.int32),
            np.ascontiguousarray(nz, np.int32))


def run_stage(st, coeffs, parts, nz, mbw, mbh, P, step=None):
    """count_probs -> num_div_denom -> encode_coefficients through one library (oracle or reference kernels)."""
    mbs = mbw * mbh
    step = step or (mbs * 25 * 16 * 3 // max(P, 1) + 4096)
    probs = np.zeros(P * 1056, np.uint32)
    denom = np.zeros(P * 1056, np.uint32)
    ctx3 = np.zeros(mbs * 25, np.uint8)
    st.count_probs(coeffs, nz, parts, probs, denom, ctx3, mbh, mbw, P)
    counted = probs.copy()
    st.num_div_denom(probs, denom, P)
    out = np.zeros(P * step, np.uint8)
    sizes = np.zeros(P, np.int32)
    st.encode_coefficients(coeffs, nz, parts, out, sizes, ctx3, probs, mbh, mbw, P, step)
    parts_bytes = [out[p * step: p * step + sizes[p]].copy() for p in range(P)]
    return dict(counts=counted, denom=denom, third_context=ctx3, probs=probs[:1056].copy(), sizes=sizes,
                partitions=parts_bytes)


def from_key_frame(W: int, H: int, seed: int, qi_min: int = 10):
    """Coefficients of a key frame (every macroblock B_PRED: MB_parts = are4x4 = 2, no Y2) from the intra oracle."""
    from intra_cases import key_case
    from oracle_lib import oracle_intra
    cur, sd = key_case(W, H, seed, qi_min)
    k = oracle_intra().intra_transform(cur, sd)
    coeffs = np.ascontiguousarray(k["MB_coeffs"], np.int16)
    parts = np.ascontiguousarray(k["MB_parts"], np.int32)
    return coeffs, parts, nz_counts(coeffs, parts)
